// How long does ONE wave take to ISSUE global loads while other waves of its CU stream LDS reads / matrix instructions?  (conv3x3_up2_g1_kernel's producers stood
// ~2300 cycles in front of eight global_load_dwordx4 for the length of the consumers' K loop: profiles/r05h_up2_producer.txt 8.)  One block per CU, 8 waves:
// waves 0-3 run the background (mode: 0 nothing, 1 ds_read_b128 stream, 2 matrix instructions, 3 both interleaved 8 : 6 as the K loop does) for the whole launch,
// wave 4 repeats { 8 x global_load_dwordx4 of scattered 16-byte pieces; s_memtime after the ISSUE of the eight; wait for the data; s_memtime } and reports the mean
// issue time and the mean time to data.  Stand-alone: hipcc --offload-arch=gfx950 -O3 tools/vmem_issue_under_lds.hip -o /tmp/viul
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f16v __attribute__((ext_vector_type(16)));
typedef float f4v __attribute__((ext_vector_type(4)));
typedef _Float16 h8v __attribute__((ext_vector_type(8)));
__global__ __launch_bounds__(512) void probe_kernel(const f4v* src, long long* out, float* sink, int mode, int trips, int probes)
{
    __shared__ __attribute__((aligned(16))) unsigned char lds[65536];
    __shared__ int done;
    const int t = threadIdx.x, wave = t >> 6, lane = t & 63;
    for (int i = t; i < 65536 / 4; i += 512) reinterpret_cast<float*>(lds)[i] = (float)i;
    if (t == 0) done = 0;
    __syncthreads();
    if (wave < 4) {
        f16v acc[2] = {(f16v)(0.0f), (f16v)(0.0f)};
        h8v f[8];
        for (int i = 0; i < 8; ++i) f[i] = (h8v)((_Float16)1.0f);
        int off = (lane * 16 + wave * 4096) & 65535;
        for (int it = 0; it < trips && !*(volatile int*)&done; ++it) {
            if (mode & 1) {
#pragma unroll
                for (int i = 0; i < 8; ++i) f[i] = *reinterpret_cast<const h8v*>(lds + ((off + i * 1024) & 65535));
                off = (off + 8192) & 65535;
            }
            if (mode & 2) {
#pragma unroll
                for (int i = 0; i < 6; ++i) acc[i & 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f[i], f[(i + 1) & 7], acc[i & 1], 0, 0, 0);
            } else if (mode & 1) { float z = 0.0f; for (int i = 0; i < 8; ++i) z += (float)f[i][0]; if (z == 12345.678f) sink[t] = z; }
        }
        if (acc[0][0] + acc[1][3] == 12345.678f) sink[t] = acc[0][1];
        return;
    }
    if (wave == 4) {
        long long ti = 0, td = 0;
        const size_t stride = 4099;                                // scattered 16-byte pieces of a 256-MB range
        size_t idx = (size_t)blockIdx.x * 65537 + lane * 131;
        float z = 0.0f;
        for (int p = 0; p < probes; ++p) {
            f4v v[8];
            const long long t0 = clock64();
#pragma unroll
            for (int i = 0; i < 8; ++i) { v[i] = src[(idx + i * stride) & ((1u << 24) - 1)]; }
            asm volatile("" ::: "memory");
            const long long t1 = clock64();                       // (s_memtime returns through lgkmcnt, not vmcnt: the loads are issued, not back)
#pragma unroll
            for (int i = 0; i < 8; ++i) z += v[i].x;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const long long t2 = clock64();
            ti += t1 - t0; td += t2 - t0;
            idx += 8 * stride + 17;
        }
        if (z == 12345.678f) sink[t] = z;
        if (lane == 0) { out[2 * blockIdx.x] = ti / probes; out[2 * blockIdx.x + 1] = td / probes; }
        __threadfence_block();
        if (lane == 0) done = 1;
    }
}
int main()
{
    f4v* src; long long* out; float* sink;
    hipMalloc(&src, (size_t)16 << 24); hipMemset(src, 0, (size_t)16 << 24); hipMalloc(&out, 256 * 16); hipMalloc(&sink, 1 << 16);
    const char* names[4] = {"nothing", "ds_read_b128 stream", "matrix instructions", "8 ds_read_b128 : 6 matrix instructions"};
    for (int mode = 0; mode < 4; ++mode) {
        hipLaunchKernelGGL(probe_kernel, dim3(256), dim3(512), 0, 0, src, out, sink, mode, 1 << 22, 200);
        hipDeviceSynchronize();
        long long h[512]; hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
        double a = 0, b = 0; for (int i = 0; i < 256; ++i) { a += h[2 * i]; b += h[2 * i + 1]; }
        printf("waves 0-3: %-40s | wave 4: 8 global_load_dwordx4 issued in %7.0f ticks, data back after %7.0f (s_memtime ticks, mean of 256 CUs x 200 probes)\n", names[mode], a / 256, b / 256);
    }
    return 0;
}
