"""Hand-written gfx950 HIP kernels and their Python bindings."""
