"""hobbyraytracer_amd — Python harness over the two C-ABI libraries of the MI355X path tracer.

The product is native: ``lib/libhrt_hip.so`` (HIP kernels for gfx950 + ``include/hrt.h``) and
``lib/libhrt_host.so`` (YAML loader, mesh import, class surface -> flat scene, film writers;
``include/hrt_host.h``), plus the ``bin/hobbyraytracer`` CLI.  This package only binds those
libraries with ctypes so that ``tests/``, ``bench.py`` and ``__graft_entry__.py`` can drive them.
There is no Python (or any CPU) implementation of the render path here: importing
:mod:`hobbyraytracer_amd.api` raises if the libraries have not been built.
"""

__all__ = ["api"]
