from .equi2pers_v3 import equi2pers  # noqa: F401
from .pers2equi_v3 import pers2equi, pers2equi_conf  # noqa: F401
