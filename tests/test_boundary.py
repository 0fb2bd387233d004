"""Drop-in boundary contract items of SURVEY.md 8(b) that had no test in round 1:
  (a) torch.nn.DataParallel(model) on device  (/root/reference/CODON_X16/test.py:52-60)
  (b) re-entrancy: host threads on their own streams (DataParallel runs replicas on Python threads; the C ABI
      promises "callable concurrently from multiple host threads / streams", include/codon_hip.h)
  (c) a C++ host with no Python: examples/host_conv.cpp compiled with hipcc against include/codon_hip.h and
      linked to libcodon_hip.so, compared bit for bit with the ctypes path
  (d) the loaded .so is the one built from the sources in the tree (stale-binary guard)."""
import os
import shutil
import subprocess
import threading

import numpy as np
import pytest
import torch

from oracle import codon_oracle as orc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _model(sd, variant="x4"):
    from codon_amd import CODONNet, CODONNet16
    m = (CODONNet16 if variant == "x16" else CODONNet)()
    m.load_state_dict(sd, strict=True)
    return m.cuda().eval()


def _rand(shape, seed):
    return torch.from_numpy(np.random.default_rng(seed).random(shape, dtype=np.float32))


@pytest.mark.gpu
def test_dataparallel_wrap_equals_bare_module_and_state_dict_prefix():
    from codon_amd import CODONNet16, strip_module_prefix
    sd = orc.he_state("x16", seed=60)
    bare = _model(sd, "x16")
    dp = torch.nn.DataParallel(_model(sd, "x16"))               # CODON_X16/test.py:52
    x, y = _rand((4, 1, 40, 56), 1).cuda(), _rand((4, 1, 40, 56), 2).cuda()
    with torch.no_grad():
        ref = bare(x, y)
        assert torch.equal(dp(x, y), ref)
    keys = list(dp.state_dict().keys())
    assert all(k.startswith("module.") for k in keys) and len(keys) == 44
    fresh = CODONNet16()
    fresh.load_state_dict(strip_module_prefix(dp.state_dict()), strict=True)      # test.py:60 path
    assert all(torch.equal(a.cpu(), b) for a, b in zip(dp.module.state_dict().values(), fresh.state_dict().values()))
    # the reference loads INTO the wrapped model with prefixed keys
    dp2 = torch.nn.DataParallel(CODONNet16().cuda())
    dp2.load_state_dict(dp.state_dict(), strict=True)
    with torch.no_grad():
        assert torch.equal(dp2(x, y), ref)


@pytest.mark.gpu
def test_dataparallel_replicas_on_threads():
    """Two replicas (replicate + scatter + parallel_apply on Python threads + gather), both on the one GPU of the
    test box: exercises _replicate_for_data_parallel and the thread re-entrancy DataParallel relies on."""
    from torch.nn.parallel import parallel_apply, replicate
    sd = orc.he_state("x4", seed=61)
    m = _model(sd)
    x, y = _rand((4, 1, 33, 47), 3).cuda(), _rand((4, 1, 33, 47), 4).cuda()
    with torch.no_grad():
        ref = m(x, y)
        reps = replicate(m, [0, 0])
        assert reps[0] is not m and reps[0]._pack_cache == {} and reps[1]._pack_cache is not reps[0]._pack_cache
        outs = parallel_apply(reps, [(x[:2].contiguous(), y[:2].contiguous()), (x[2:].contiguous(), y[2:].contiguous())],
                              devices=[0, 0])
    assert torch.equal(torch.cat(outs), ref)


@pytest.mark.gpu
def test_two_host_threads_two_streams_match_single_thread():
    sd = orc.he_state("x4", seed=62)
    m = _model(sd)
    ins = [(_rand((2, 1, 48, 64), 10 + i).cuda(), _rand((2, 1, 48, 64), 20 + i).cuda()) for i in range(2)]
    with torch.no_grad():
        refs = [m(a, b) for a, b in ins]            # also packs the weights once, before the threads start
    torch.cuda.synchronize()
    outs, errs = [None, None], []

    def work(i):
        try:
            st = torch.cuda.Stream()
            with torch.no_grad(), torch.cuda.stream(st):
                for _ in range(20):
                    o = m(*ins[i])
                st.synchronize()
            outs[i] = o
        except Exception as e:  # noqa: BLE001
            errs.append(e)

    ts = [threading.Thread(target=work, args=(i,)) for i in range(2)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errs, errs
    for o, r in zip(outs, refs):
        assert torch.equal(o, r)


@pytest.mark.gpu
def test_error_string_is_per_thread():
    """codon_last_error_string() is thread-local: a failing call on one thread does not clobber another's."""
    import ctypes as C
    from codon_amd import _lib as L
    lib = L.load()
    lib.codon_last_error_string.restype = C.c_char_p
    got = {}

    def bad(ks, key):
        d = L.ConvDesc(1, 8, 8, 64, 64, ks, 64, 0, 64, 0, 0, 0, 0, L.F32)
        rc = lib.codon_conv2d_fwd(C.byref(d), None, None, None, None, None)
        got[key] = (rc, lib.codon_last_error_string().decode())

    t = threading.Thread(target=bad, args=(4, "t"))
    bad(7, "main")
    t.start(); t.join()
    assert got["main"][0] < 0 and got["t"][0] < 0
    # the main thread's message is still its own after the other thread failed with a different one
    assert lib.codon_last_error_string().decode() == got["main"][1]


def _lcg(n, salt):
    i = np.arange(n, dtype=np.uint64)
    u = (i * np.uint64(2654435761) + np.uint64(salt)) & np.uint64(0xffffffff)
    return (((u >> np.uint64(8)) & np.uint64(0xffff)).astype(np.float32) / np.float32(65536.0) - np.float32(0.5))


@pytest.mark.gpu
def test_cpp_host_program_matches_ctypes_path(tmp_path):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not on this box")
    exe = str(tmp_path / "host_conv")
    subprocess.run([hipcc, "--offload-arch=gfx950", "-O2", "-I", os.path.join(ROOT, "include"),
                    os.path.join(ROOT, "examples", "host_conv.cpp"), "-L", os.path.join(ROOT, "codon_amd"),
                    "-lcodon_hip", "-Wl,-rpath," + os.path.join(ROOT, "codon_amd"), "-o", exe],
                   check=True, timeout=600)
    outp = str(tmp_path / "y.f32")
    r = subprocess.run([exe, outp], check=True, timeout=300, capture_output=True, text=True)
    assert "host_conv:" in r.stdout
    B, H, W, Cn, K = 2, 19, 45, 128, 5
    y_host = np.fromfile(outp, dtype=np.float32).reshape(B, Cn, H, W)
    from codon_amd import _lib as L
    from codon_amd import ops
    x = torch.from_numpy(_lcg(B * Cn * H * W, 17).reshape(B, Cn, H, W)).cuda()
    w = torch.from_numpy((_lcg(Cn * Cn * K * K, 99) * np.float32(0.05)).reshape(Cn, Cn, K, K)).cuda()
    y = torch.empty_like(x)
    ops.conv2d(ops.Slice(x), ops.packed_weight(w, L.PACK_FWD), ops.Slice(y), K, relu=True)
    assert np.array_equal(y.cpu().numpy(), y_host)
    # and both equal the textbook op
    ref = torch.relu(torch.nn.functional.conv2d(x.cpu().double(), w.cpu().double(), padding=2))
    assert float((y.cpu().double() - ref).abs().max()) < 1e-5


def test_library_is_built_from_the_sources_in_the_tree():
    from codon_amd import _lib as L
    info = L.build_info()
    assert info["source_hash_now"] == info["source_hash_built"], (
        "codon_amd/libcodon_hip.so was built from different csrc/ + include/ sources than the ones in the tree: "
        "run `make -C codon_amd/csrc`")
