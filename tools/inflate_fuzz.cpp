// Differential fuzz of csrc/omni_inflate.h against libz under the sanitizers (exact-size heap buffers: any access outside them is reported):
//   g++ -O1 -g -fsanitize=address,undefined -fno-sanitize-recover=all -std=c++20 tools/inflate_fuzz.cpp -lz -o /tmp/inflate_fuzz && /tmp/inflate_fuzz <seed> <iterations>
// random payloads of four kinds at zlib levels 0-9, intact / truncated / 1-3 flipped bits, destination exact / short / long: accept and refuse as libz, same bytes.
// (round 5: 4 seeds x 30 000 iterations clean; round 6: payloads to 4 MB and a photo-like generator, 2 seeds x 4 000 clean)
#include "../omnifusion_amd/csrc/omni_inflate.h"
#include <zlib.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <random>
int main(int argc, char** argv)
{
    std::mt19937_64 rng(argc > 1 ? atoi(argv[1]) : 1);
    long ok = 0, bad = 0, agree = 0;
    const int iters = argc > 2 ? atoi(argv[2]) : 20000;
    for (int it = 0; it < iters; ++it) {
        // sizes: mostly small (every boundary of the fast loop is near), 1 in 8 up to 400 KB (match distances to 32 KiB, long codes behind the second-level
        // tables, long runs inside the fast loop: ADVICE r5), 1 in 64 a few MB (a real panorama inflates to 1.5-25 MB)
        const unsigned pick = (unsigned)(rng() % 64);
        const size_t n = pick == 0 ? rng() % 4000000 + 1 : (pick < 9 ? rng() % 400000 + 1 : rng() % 5000 + 1);
        std::vector<unsigned char> d(n);
        const int mode = rng() % 5;
        unsigned char v = 0;
        const size_t rowlen = 3 * (64 + rng() % 2000);           // mode 4: photo-like — a smooth gradient + noise, rows that resemble the row above
        for (size_t i = 0; i < n; ++i) {
            if (mode == 0) d[i] = (unsigned char)rng();
            else if (mode == 4) d[i] = (unsigned char)((i >= rowlen ? d[i - rowlen] : (unsigned char)(i / 7)) + (int)(rng() % 9) - 4 + ((rng() & 63) == 0 ? (int)(rng() % 64) : 0));
            else if (mode == 1) d[i] = (unsigned char)(rng() % 4);
            else if (mode == 2) { v = (unsigned char)(v + (int)(rng() % 5) - 2); d[i] = v; }
            else d[i] = (unsigned char)((i % 7) ? 0 : (rng() & 1 ? 255 : 1));
        }
        uLongf zn = compressBound(n);
        std::vector<unsigned char> z(zn);
        const int lvl = (int)(rng() % 10);
        compress2(z.data(), &zn, d.data(), n, lvl);
        // exact-size heap buffers: ASAN sees any access outside them
        size_t cut = zn;
        const int mut = rng() % 4;
        if (mut == 1) cut = rng() % (zn + 1);
        unsigned char* in = (unsigned char*)malloc(cut ? cut : 1);
        memcpy(in, z.data(), cut);
        if (mut >= 2 && cut) for (int k = 0; k < 1 + (int)(rng() % 3); ++k) in[rng() % cut] ^= (unsigned char)(1u << (rng() % 8));
        size_t cap = n;
        const int cm = rng() % 4;
        if (cm == 1) cap = rng() % (n + 1); else if (cm == 2) cap = n + rng() % 300;
        unsigned char* out = (unsigned char*)malloc(cap ? cap : 1);
        size_t used = 0, prod = 0;
        const int rc = omni_inflate::zlib_decompress(in, cut, out, cap, &used, &prod);
        if (prod > cap || used > cut) { printf("BOUNDS it %d\n", it); return 1; }
        // reference
        std::vector<unsigned char> ref(cap + 1);
        z_stream zs; memset(&zs, 0, sizeof zs); inflateInit(&zs);
        zs.next_in = in; zs.avail_in = (uInt)cut; zs.next_out = ref.data(); zs.avail_out = (uInt)(cap + 1);
        const int zr = inflate(&zs, Z_FINISH);
        const bool ref_ok = zr == Z_STREAM_END && zs.total_out <= cap;
        inflateEnd(&zs);
        if ((rc == 0) != ref_ok) { printf("DISAGREE it %d rc %d zr %d total_out %lu cap %zu mut %d\n", it, rc, zr, zs.total_out, cap, mut); return 1; }
        if (rc == 0) { if (prod != zs.total_out || memcmp(out, ref.data(), prod)) { printf("DATA it %d\n", it); return 1; } ++ok; }
        else { if (memcmp(out, d.data(), 0)) return 1; ++bad; }
        ++agree;
        free(in); free(out);
    }
    printf("iterations %ld: accepted %ld refused %ld, all as libz\n", agree, ok, bad);
    return 0;
}
