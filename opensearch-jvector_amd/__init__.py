"""opensearch-jvector_amd — MI355X-native engine for the jVector GraphSearcher hot path.

The product is the C-ABI library (``lib/libjvgpu.so``, declared in ``include/jvgpu.h``) built from
``csrc/``; this Python package is plumbing around it for tests and the benchmark:

* ``binding``  — ctypes view of the C ABI (fails loudly when the HIP library is missing);
* ``builder``  — ctypes view of the write-side helper (graph / PQ construction, not the hot path);
* ``datagen``  — bit-reproducible synthetic inputs (java.util.Random LCG, splitmix64);
* ``host``     — ctypes view of the C++ mirror of the reference's reader/collector/query classes.

The directory name carries a hyphen (it is fixed by the project layout), so import it through
``__graft_entry__.load_package()``, which registers it as ``opensearch_jvector_amd``.
"""
from . import datagen  # noqa: F401  (pure numpy, always importable)

__all__ = ["datagen", "binding", "builder"]
