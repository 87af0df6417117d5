"""cat_amd: MI355X-native RNS polynomial arithmetic (NTT, key switching, rescale) behind a C ABI.

The product is cat_amd/lib/libcat_amd.so (hand-written HIP for gfx950, include/cat_amd.h).  This
package is thin plumbing for tests and bench.py: torch owns device memory and streams, ctypes calls
the C ABI.  The C++ facade with the reference's class names lives in cat_amd/host/.
"""
from .api import CatError, Context, SCHEME_BFV, SCHEME_CKKS, SCHEME_BGV, SCHEME_NONE, galois_elt_from_step  # noqa: F401
