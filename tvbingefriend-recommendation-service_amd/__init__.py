"""MI355X-native batch similarity + top-K path of the TV-show recommender.

Layout
  csrc/      HIP kernels for gfx950 and the C ABI (include/tvbf.h)
  native.py  ctypes binding of that ABI (fails loudly when the library is missing)
  engine.py  device-side operations on torch tensors (memory, streams)
  features.py  feature blocks -> one row-major fp32 matrix in HBM
  sharding.py  row sharding over ranks + all-gather of the normalised block
  ml/, services/, scripts/  drop-in mirrors of the reference's Python surface
"""

__version__ = "0.1.0"
