"""snx -- Python binding of libsnx.so, the MI355X-native (gfx950) HIP implementation of the
SPLADE-ModernBERT training hot path.  PyTorch is used for device memory, streams and
torch.distributed only; every hot-path op is a hand-written HIP kernel behind the C ABI declared
in include/snx.h."""
from ._lib import SnxError, SnxLibraryError, config, configure, fn, lib, verify_exports  # noqa: F401
