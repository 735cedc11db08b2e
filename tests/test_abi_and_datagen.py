"""The C-ABI library loads and exports every symbol include/jvgpu.h declares (no compute calls: no
GPU here); the product path fails loudly without a device; input generators are reproducible."""
import ctypes
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol(pkg):
    lib = pkg.binding.load_library()
    with open(os.path.join(ROOT, "include", "jvgpu.h")) as f:
        hdr = f.read()
    declared = sorted(set(re.findall(r"\b(jv_[a-z_]+)\s*\(", hdr)))
    assert declared, "no declarations found"
    assert sorted(pkg.binding.ABI_SYMBOLS) == declared, "binding.ABI_SYMBOLS must list exactly the header's functions"
    for sym in declared:
        assert hasattr(lib, sym), f"{sym} is declared in include/jvgpu.h but not exported"
    assert lib.jv_abi_version() == 4


def test_desc_struct_layout_matches_header(pkg):
    # 64-bit layout of jv_index_desc: 10 x 4 B + pad, pointers 8-aligned
    assert ctypes.sizeof(pkg.binding.JvIndexDesc) == 160
    assert pkg.binding.JvIndexDesc.vectors.offset == 48
    assert pkg.binding.JvIndexDesc.ord2doc.offset == 104
    assert pkg.binding.JvIndexDesc.nvq_M.offset == 120 and pkg.binding.JvIndexDesc.nvq_global_mean.offset == 152


@pytest.mark.skipif(os.path.exists("/dev/kfd"), reason="GPU present")
def test_no_cpu_fallback_without_gpu(pkg):
    """Without a HIP device the product path must fail loudly (JV_EDEVICE), never compute on the CPU."""
    b = pkg.binding
    ix = b.IndexData(vectors=np.zeros((4, 2), np.float32), adj=np.full((4, 2), -1, np.int32), entry_node=0)
    with pytest.raises(b.JvError) as ei:
        b.GpuIndex(ix)
    assert ei.value.code == b.JV_EDEVICE


def test_argument_validation_without_gpu(pkg):
    b = pkg.binding
    lib = b.load_library()
    h = ctypes.c_void_p()
    assert lib.jv_index_create(None, ctypes.byref(h)) == b.JV_EINVAL
    desc, keep = b.make_desc(b.IndexData(vectors=np.zeros((4, 2), np.float32), adj=np.full((4, 2), -1, np.int32), entry_node=0))
    desc.similarity = 7  # VectorSimilarityMapper.ordToDistFunc -> IllegalArgumentException
    assert lib.jv_index_create(ctypes.byref(desc), ctypes.byref(h)) == b.JV_EINVAL
    assert b"similarity" in lib.jv_last_error()
    desc.similarity = 0
    desc.struct_size = 8
    assert lib.jv_index_create(ctypes.byref(desc), ctypes.byref(h)) == b.JV_EINVAL
    assert lib.jv_set_option(b"no_such_option", 1) == b.JV_EINVAL
    lib.jv_index_destroy(None)  # NULL is a no-op


def test_java_random_known_answers(pkg):
    """java.util.Random known values: new Random(42).nextFloat() = 0.7275637, Random(0) = 0.73096776."""
    dg = pkg.datagen
    assert dg.java_random_floats(42, 1)[0] == np.float32(0.7275637)
    assert dg.java_random_floats(0, 1)[0] == np.float32(0.73096776)
    x = dg.java_random_floats(1, 70000)
    y = dg.java_random_floats(1, 10)
    assert np.array_equal(x[:10], y) and 0 <= x.min() and x.max() < 1
    # scalar restatement
    s = (1 ^ 0x5DEECE66D) & ((1 << 48) - 1)
    for i in range(70000):
        s = (s * 0x5DEECE66D + 0xB) & ((1 << 48) - 1)
    assert x[-1] == np.float32((s >> 24) / float(1 << 24))


def test_splitmix_generator_is_counter_based(pkg):
    dg = pkg.datagen
    a = dg.splitmix_uniform(42, 100, 16)
    b = dg.splitmix_uniform(42, 40, 16, row_offset=60)
    assert np.array_equal(a[60:], b)  # shards can generate their own doc range independently
    assert a.dtype == np.float32 and 0 <= a.min() and a.max() < 1
    assert abs(a.mean() - 0.5) < 0.05


def test_bench_uniform_distribution_equals_the_numpy_generator(pkg):
    """bench.py's `uniformA` rows (SURVEY 8(d) distribution A, generated in HBM with torch int64 arithmetic) are bit-equal to
    datagen.splitmix_uniform — the same counter-based generator the tests and the oracle's fixtures use"""
    import os
    import sys
    import torch
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    for seed, n, d, off in [(42, 300, 768, 0), (43, 17, 128, 123456789), (7, 5, 3, 2 ** 33)]:
        got = bench.splitmix_uniform_torch(torch, seed, n, d, off, torch.device("cpu")).numpy()
        want = pkg.datagen.splitmix_uniform(seed, n, d, off)
        assert got.dtype == np.float32 and np.array_equal(got.view(np.uint32), want.view(np.uint32)), (seed, n, d, off)
    assert "uniformA" in bench.DISTS


def test_hardware_probes_cross_compile(tmp_path):
    """tools/lds_residency.hip and tools/hwq_probe.hip (the measurements behind the rung sizes and the query servers' stream
    priorities) must keep compiling for gfx950 — hipcc cross-compiles without a GPU."""
    import os
    import shutil
    import subprocess
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for src in ("lds_residency.hip", "hwq_probe.hip", "start_overlap_probe.hip", "wave_placement_probe.hip"):   # (+ round 4's probes)
        out = tmp_path / (src + ".o")
        r = subprocess.run([hipcc, "--offload-arch=gfx950", "-O2", "-c", "-o", str(out), os.path.join(root, "tools", src)],
                           capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
