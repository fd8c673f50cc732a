"""Version of the MI355X build of the S^3 hot path (independent of the reference's own version number).

The C ABI carries its own integer, ``s3_abi_version()`` (include/s3hip.h); it changes only when an existing entry point
changes its meaning or signature."""

VERSION_INFO = (0, 1, 0)
__version__ = ".".join(str(part) for part in VERSION_INFO)
