"""omnifusion_amd — MI355X-native (gfx950) OmniFusion `equi_pers` hot path.

Host-side mirror of the reference's Python interface over the C-ABI library
libomnifusion_hip.so (include/omnifusion.h):

    from omnifusion_amd.equi_pers.equi2pers_v3 import equi2pers     # equi_pers/equi2pers_v3.py:20
    from omnifusion_amd.equi_pers.pers2equi_v3 import pers2equi     # equi_pers/pers2equi_v3.py:16
    from omnifusion_amd.model.spherical_model import spherical_fusion            # model/spherical_model.py:190
    from omnifusion_amd.model.spherical_model_iterative import spherical_fusion  # ..._iterative.py:253
"""
__version__ = "0.1.0"
