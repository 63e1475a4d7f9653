"""objgan_hip: MI355X (gfx950) native kernels for the Obj-GAN image_generation hot path.

`ops` holds the autograd bindings, `_lib` the ctypes loader of libobjgan_hip.so (C-ABI in
include/objgan_hip.h), `build` the hipcc build recipe.
"""
from . import _lib, build  # noqa: F401
from ._lib import ObjganHipError  # noqa: F401
