"""CPU-only checks of the drop-in boundary and the host logic:
  * libyfv2.so loads and exports every function include/yfv2.h declares
  * argument / state errors come back as negative codes with a message (no GPU needed)
  * the Python Detector has exactly the reference state_dict key set
  * batch sharding + the world_size-2 all-gather of padded detections (gloo)
No compute call is made here: without a GPU yfv2_create must refuse, loudly.
"""
import ctypes as C
import os
import re
import subprocess
import sys
import textwrap

import numpy as np
import pytest
import torch

from conftest import GOLDEN, REPO


def _lib():
    from yolo_fastestv2_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    return _lib


def test_library_exports_every_declared_symbol():
    _lib_mod = _lib()
    hdr = open(os.path.join(REPO, "include", "yfv2.h")).read()
    declared = set(re.findall(r"YFV2_API\s+[\w\s\*]+?\b(yfv2_\w+)\s*\(", hdr))
    assert len(declared) >= 15, declared
    assert declared == set(_lib_mod._PROTOTYPES), declared ^ set(_lib_mod._PROTOTYPES)
    raw = C.CDLL(_lib_mod.LIB_PATH)
    for name in declared:
        assert hasattr(raw, name), "libyfv2.so does not export %s" % name
    assert _lib_mod.lib().yfv2_abi_version() == _lib_mod.ABI_VERSION == 6
    # nothing else leaks out of the library's namespace
    syms = subprocess.run(["nm", "-D", "--defined-only", _lib_mod.LIB_PATH], capture_output=True, text=True).stdout
    exported = {l.split()[-1] for l in syms.splitlines() if " T " in l}
    assert {s for s in exported if s.startswith("yfv2_")} == declared


def test_host_checksum_that_guards_the_nms_cache():
    """non_max_suppression reuses the device copy handel_preds left on its result only while the host tensor is unedited: the checksum
    it compares changes under a torch write AND under a numpy-side write (which does not bump the tensor's version); tensors beyond
    8 MB are never cached."""
    from yolo_fastestv2_amd.utils.utils import _host_checksum
    t = torch.rand(3, 1815, 85)
    c0 = _host_checksum(t)
    assert c0 is not None and c0 == _host_checksum(t)
    v = t._version
    t.numpy()[1, 7, 4] = 0.25
    assert t._version == v and _host_checksum(t) != c0
    c1 = _host_checksum(t)
    t[2, 0, 0] += 1.0
    assert _host_checksum(t) != c1
    assert _host_checksum(t.clone()) != _host_checksum(t)          # another buffer is another tensor, whatever it holds
    assert _host_checksum(torch.zeros(16, 1815, 85)) is None        # 9.9 MB: not cached, uploaded again
    assert _host_checksum(t[:, ::2]) is None                        # a strided view is not what handel_preds returned


def test_errors_without_gpu_are_codes_not_crashes():
    m = _lib()
    L = m.lib()
    cfg = m.Config()
    cfg.classes, cfg.anchor_num, cfg.height, cfg.width, cfg.max_batch, cfg.device = 80, 3, 352, 352, 1, 0
    h = C.c_void_p()
    assert L.yfv2_create(None, C.byref(cfg)) == m.ERR_ARG
    bad = m.Config(); bad.classes, bad.anchor_num, bad.height, bad.width, bad.max_batch = 80, 5, 352, 352, 1
    assert L.yfv2_create(C.byref(h), C.byref(bad)) == m.ERR_CONFIG and "anchor_num" in m.last_error()
    bad.anchor_num, bad.height = 3, 350
    assert L.yfv2_create(C.byref(h), C.byref(bad)) == m.ERR_CONFIG
    if not torch.cuda.is_available():
        rc = L.yfv2_create(C.byref(h), C.byref(cfg))
        assert rc == m.ERR_DEVICE and not h.value
        assert "no CPU fallback" in m.last_error()
    assert L.yfv2_forward(None, None, 1, None, None) == m.ERR_ARG
    assert L.yfv2_num_rows(None) == 0 and L.yfv2_num_stages(None) == 0
    L.yfv2_destroy(None)  # no-op


def test_python_surface_refuses_cpu():
    import yolo_fastestv2_amd as yfv2
    with pytest.raises(RuntimeError):
        yfv2.Engine("cpu")
    m = yfv2.Detector(80, 3, True).eval()
    with pytest.raises(RuntimeError):
        m(torch.rand(1, 3, 352, 352))
    with pytest.raises(RuntimeError):
        yfv2.handel_preds([torch.zeros(1)] * 6, {"height": 352, "width": 352, "anchor_num": 3, "anchors": [0] * 12}, "cpu")


def test_product_never_imports_the_oracle():
    pkg = os.path.join(REPO, "yolo_fastestv2_amd")
    for root, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(root, f)).read()
                assert "oracle" not in src.replace("no oracle", ""), "%s mentions the oracle" % f


def test_detector_state_dict_keys_match_reference_checkpoint():
    import yolo_fastestv2_amd as yfv2
    z = np.load(os.path.join(GOLDEN, "weights_coco.npz"))
    m = yfv2.Detector(80, 3, True)
    sd = m.state_dict()
    assert list(sd.keys()) == list(z.files)  # same keys, same order as the reference module tree
    for k in z.files:
        assert tuple(sd[k].shape) == z[k].shape, k
    assert sum(p.numel() for p in m.parameters()) == 243095
    res = m.load_state_dict({k: torch.from_numpy(z[k]) for k in z.files})
    assert not res.missing_keys and not res.unexpected_keys
    # optimizer / device plumbing the reference scripts rely on
    torch.optim.SGD(m.parameters(), lr=0.1)
    assert m.train().training and not m.eval().training


def test_load_datafile(tmp_path):
    import yolo_fastestv2_amd as yfv2
    p = tmp_path / "x.data"
    p.write_text("[name]\nmodel_name=coco\n\n[model-configure]\npre_weights=None\nclasses=80\nwidth=352\nheight=352\n"
                 "anchor_num=3\nanchors=12.64,19.39, 37.88,51.48, 55.71,138.31, 126.91,78.23, 131.57,214.55, 279.92,258.87\n"
                 "learning_rate=0.001\nsteps=150,250\n")
    cfg = yfv2.load_datafile(str(p))
    assert cfg["classes"] == 80 and cfg["height"] == 352 and cfg["pre_weights"] == "None"
    assert cfg["anchors"][:2] == [12.64, 19.39] and len(cfg["anchors"]) == 12 and cfg["steps"] == [150.0, 250.0]
    z = np.load(os.path.join(GOLDEN, "cfg_coco.npz"))
    assert cfg["anchors"] == [float(a) for a in z["anchors"]]


def test_shard_range_partitions():
    from yolo_fastestv2_amd import shard_range
    for n, w in ((2048, 8), (256, 1), (10, 4), (3, 8)):
        spans = [shard_range(n, r, w) for r in range(w)]
        assert spans[0][0] == 0 and spans[-1][1] == n
        assert all(spans[i][1] == spans[i + 1][0] for i in range(w - 1))
        sizes = [b - a for a, b in spans]
        assert max(sizes) - min(sizes) <= 1
    assert shard_range(2048, 3, 8) == (768, 1024)


def test_gather_detections_world_size_2_gloo(tmp_path):
    """N>1 path on CPU: two processes, gloo, each contributes its shard of padded detections."""
    script = tmp_path / "w.py"
    script.write_text(textwrap.dedent("""
        import os, sys, torch, torch.distributed as dist
        sys.path.insert(0, %r)
        from yolo_fastestv2_amd import gather_detections, shard_range
        rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
        dist.init_process_group("gloo", rank=rank, world_size=world)
        B = 6
        lo, hi = shard_range(B, rank, world)
        g = torch.Generator().manual_seed(0)
        dets_all = torch.rand(B, 300, 6, generator=g); idx_all = torch.randint(0, 1815, (B, 300), generator=g, dtype=torch.int32)
        cnt_all = torch.randint(0, 300, (B,), generator=g, dtype=torch.int32)
        d, i, c = gather_detections(dets_all[lo:hi].clone(), idx_all[lo:hi].clone(), cnt_all[lo:hi].clone())
        assert torch.equal(d, dets_all) and torch.equal(i, idx_all) and torch.equal(c, cnt_all), rank
        dist.barrier(); dist.destroy_process_group()
        print("rank", rank, "ok")
    """ % REPO))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", WORLD_SIZE="2")
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r)), stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT, text=True) for r in range(2)]
    outs = [p.communicate(timeout=120)[0] for p in procs]
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0 and "rank %d ok" % r in o, o


def test_packed_and_async_gather_world_size_2_gloo(tmp_path):
    """The bench's N>1 step on CPU: buffers from packed_det_buffers (what Engine.new_det_buffers returns) travel as ONE
    collective into a reusable flat receive buffer; async_op=True returns a handle whose wait() yields the same tensors;
    two buffer sets in flight, as bench.py keeps them."""
    script = tmp_path / "w.py"
    script.write_text(textwrap.dedent("""
        import os, sys, torch, torch.distributed as dist
        sys.path.insert(0, %r)
        from yolo_fastestv2_amd import gather_detections, shard_range
        from yolo_fastestv2_amd import sharded
        rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
        dist.init_process_group("gloo", rank=rank, world_size=world)
        B = 3                                                            # per rank
        calls = []
        real = dist.all_gather_into_tensor
        def counting(*a, **k):
            calls.append(1); return real(*a, **k)
        dist.all_gather_into_tensor = counting
        g = torch.Generator().manual_seed(0)
        sets, recv, works, want = [], [], [], []
        for step in range(4):
            dets_all = torch.rand(world * B, 300, 6, generator=g); idx_all = torch.randint(0, 1815, (world * B, 300), generator=g, dtype=torch.int32)
            cnt_all = torch.randint(0, 300, (world * B,), generator=g, dtype=torch.int32)
            if step < 2:
                sets.append(sharded.packed_det_buffers(B, "cpu")); recv.append(torch.empty(world * B * 2101)); works.append(None)
            j = step %% 2
            if works[j] is not None:
                if step == 2: works[j].wait_host()                       # host-side completion, then the ordinary wait
                d, i, c = works[j].wait()
                assert torch.equal(d, want[j][0]) and torch.equal(i, want[j][1]) and torch.equal(c, want[j][2]), (rank, step)
            lo, hi = rank * B, (rank + 1) * B
            sets[j][0].copy_(dets_all[lo:hi]); sets[j][1].copy_(idx_all[lo:hi]); sets[j][2].copy_(cnt_all[lo:hi])
            works[j] = gather_detections(*sets[j], async_op=True, out=recv[j])
            if step < 2: want.append(None)
            want[j] = (dets_all, idx_all, cnt_all)
        for j in range(2):
            assert works[j].wait(unpack=False) is None
            for r, (d, i, c) in enumerate(sharded.rank_views(recv[j], world, B)):          # zero-copy per-rank views
                assert d.data_ptr() >= recv[j].data_ptr() and torch.equal(d, want[j][0][r * B:(r + 1) * B])
                assert torch.equal(i, want[j][1][r * B:(r + 1) * B]) and torch.equal(c, want[j][2][r * B:(r + 1) * B])
            d, i, c = works[j].wait()
            assert torch.equal(d, want[j][0]) and torch.equal(i, want[j][1]) and torch.equal(c, want[j][2]), (rank, "tail", j)
        assert len(calls) == 4, calls                                    # one collective per step
        d, i, c = gather_detections(*sets[0])                            # synchronous form, own receive buffer
        assert torch.equal(d, want[0][0]) and torch.equal(i, want[0][1]) and torch.equal(c, want[0][2])
        assert len(calls) == 5
        d, i, c = gather_detections(sets[0][0].clone(), sets[0][1].clone(), sets[0][2].clone())   # not packed: three collectives
        assert torch.equal(d, want[0][0]) and len(calls) == 8
        dist.barrier(); dist.destroy_process_group()
        print("rank", rank, "ok")
    """ % REPO))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29541", WORLD_SIZE="2")
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r)), stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT, text=True) for r in range(2)]
    outs = [p.communicate(timeout=120)[0] for p in procs]
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0 and "rank %d ok" % r in o, o


def test_detect_sharded_world_size_2_gloo_reassembles_the_single_rank_result(tmp_path):
    """SURVEY.md 8(e) end to end on CPU: the batch is split with shard_range, every rank runs `detect` on ITS images only
    (a stand-in engine whose detections are a deterministic function of each image, so a wrong split, order or padding
    shows), detect_sharded gathers - every rank must hold exactly what one rank computes on the whole batch.  Also the
    debug mode that gathers the decoded tensor."""
    script = tmp_path / "w.py"
    script.write_text(textwrap.dedent("""
        import os, sys, torch, torch.distributed as dist
        sys.path.insert(0, %r)
        from yolo_fastestv2_amd import detect_sharded, gather_decoded, shard_range

        class FakeEngine:                     # same call shape as Engine.detect
            def detect(self, x, conf_thres, iou_thres, out=None):
                B = x.shape[0]
                s = x.reshape(B, -1).double().sum(1)                      # a fingerprint of each image
                cnt = (s * 7).long().remainder(300).to(torch.int32)
                k = torch.arange(300)[None, :, None].double()
                dets = ((s[:, None, None] + k) * torch.arange(1, 7)[None, None, :]).float() * float(conf_thres + iou_thres)
                idx = ((s[:, None] * 13).long() + torch.arange(300)[None]).remainder(1815).to(torch.int32)
                live = torch.arange(300)[None] < cnt[:, None]
                return dets * live[..., None], idx * live, cnt

        rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
        dist.init_process_group("gloo", rank=rank, world_size=world)
        g = torch.Generator().manual_seed(5)
        x_all = torch.rand(8, 3, 32, 32, generator=g)                    # every rank builds the same global batch
        lo, hi = shard_range(x_all.shape[0], rank, world)
        eng = FakeEngine()
        d, i, c = detect_sharded(eng, x_all[lo:hi], 0.3, 0.4)
        D, I, C = eng.detect(x_all, 0.3, 0.4)
        assert d.shape == D.shape and torch.equal(d, D) and torch.equal(i, I) and torch.equal(c, C), rank
        dec_all = torch.rand(8, 60, 85, generator=g)
        assert torch.equal(gather_decoded(dec_all[lo:hi].clone()), dec_all), rank
        dist.barrier(); dist.destroy_process_group()
        print("rank", rank, "ok")
    """ % REPO))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29537", WORLD_SIZE="2")
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r)), stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT, text=True) for r in range(2)]
    outs = [p.communicate(timeout=120)[0] for p in procs]
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0 and "rank %d ok" % r in o, o


def test_detect_sharded_world_size_8_gloo_packed_layout(tmp_path):
    """BASELINE configs[3]'s geometry on CPU: EIGHT ranks (one per GPU of a node), contiguous shards of a global batch, every
    rank's result in the packed buffers `Engine.new_det_buffers` hands out, ONE all_gather_into_tensor per step, issued
    asynchronously with a reusable receive buffer exactly as bench.py --gpus 8 does.  Asserted: the receive buffer's layout
    [rank][dets | idx | cnt] word for word (what `rank_views` promises), the unpacked result == what one rank computes on the
    whole batch, one collective per step."""
    script = tmp_path / "w8.py"
    script.write_text(textwrap.dedent("""
        import os, sys, torch, torch.distributed as dist
        sys.path.insert(0, %r)
        from yolo_fastestv2_amd import gather_detections, shard_range, sharded

        class FakeEngine:                     # same call shape as Engine.detect / Engine.new_det_buffers
            def new_det_buffers(self, B):
                return sharded.packed_det_buffers(B, "cpu")
            def detect(self, x, conf_thres, iou_thres, out=None):
                B = x.shape[0]
                dets, idx, cnt = out if out is not None else self.new_det_buffers(B)
                s = x.reshape(B, -1).double().sum(1)
                cnt.copy_((s * 7).long().remainder(301).to(torch.int32))
                k = torch.arange(300)[None, :, None].double()
                dets.copy_(((s[:, None, None] + k) * torch.arange(1, 7)[None, None, :]).float())
                idx.copy_(((s[:, None] * 13).long() + torch.arange(300)[None]).remainder(1815).to(torch.int32))
                return dets, idx, cnt

        rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
        dist.init_process_group("gloo", rank=rank, world_size=world)
        calls = [0]
        real = dist.all_gather_into_tensor
        def counted(*a, **k):
            calls[0] += 1
            return real(*a, **k)
        dist.all_gather_into_tensor = counted
        Bl = 4                                                             # images per rank (256 on the real node)
        g = torch.Generator().manual_seed(11)
        x_all = torch.rand(world * Bl, 3, 16, 16, generator=g)            # every rank builds the same global batch
        lo, hi = shard_range(x_all.shape[0], rank, world)
        assert (lo, hi) == (rank * Bl, (rank + 1) * Bl)
        eng = FakeEngine()
        recv = torch.empty(world * Bl * 2101)
        for step in range(2):                                              # the receive buffer is reused step after step
            d, i, c = eng.detect(x_all[lo:hi] + step, 0.3, 0.4, out=eng.new_det_buffers(Bl))
            work = gather_detections(d, i, c, async_op=True, out=recv)
            work.wait_host()
            D, I, C = eng.detect(x_all + step, 0.3, 0.4)
            r = recv.view(world, Bl * 2101)
            for q in range(world):                                         # [rank q][dets (Bl,300,6) | idx (Bl,300) | cnt (Bl)]
                sl = slice(q * Bl, (q + 1) * Bl)
                assert torch.equal(r[q, :Bl * 1800].view(Bl, 300, 6), D[sl]), (rank, q)
                assert torch.equal(r[q, Bl * 1800:Bl * 2100].view(torch.int32).view(Bl, 300), I[sl]), (rank, q)
                assert torch.equal(r[q, Bl * 2100:].view(torch.int32), C[sl]), (rank, q)
            for q, (vd, vi, vc) in enumerate(sharded.rank_views(recv, world, Bl)):
                assert vd.data_ptr() == recv.data_ptr() + 4 * q * Bl * 2101
            gd, gi, gc = work.wait()
            assert torch.equal(gd, D) and torch.equal(gi, I) and torch.equal(gc, C), rank
        assert calls[0] == 2, calls                                        # ONE collective per step
        dist.barrier(); dist.destroy_process_group()
        print("rank", rank, "ok")
    """ % REPO))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29541", WORLD_SIZE="8", OMP_NUM_THREADS="1")
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r)), stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT, text=True) for r in range(8)]
    outs = [p.communicate(timeout=300)[0] for p in procs]
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0 and "rank %d ok" % r in o, o


def test_gather_is_identity_without_process_group():
    from yolo_fastestv2_amd import gather_detections
    d, i, c = torch.rand(2, 300, 6), torch.zeros(2, 300, dtype=torch.int32), torch.zeros(2, dtype=torch.int32)
    out = gather_detections(d, i, c)
    assert out[0] is d and out[1] is i and out[2] is c


def test_export_weights_container_round_trip(tmp_path):
    """The flat container include/yfv2.hpp's Detector::loadModel reads: every floating tensor, in order, bit-exact;
    integer buffers (num_batches_tracked) skipped."""
    import struct

    import numpy as np

    import yolo_fastestv2_amd as yfv2

    w = yfv2.random_state_dict(seed=3)
    path = str(tmp_path / "w.yfv2w")
    n = yfv2.export_weights(w, path)
    floats = [(k, v) for k, v in w.items() if v.is_floating_point()]
    assert n == len(floats) and n < len(w)
    buf = open(path, "rb").read()
    assert buf[:8] == b"YFV2W1\0\0" and struct.unpack_from("<i", buf, 8)[0] == n
    off = 12
    for k, v in floats:
        (ln,) = struct.unpack_from("<i", buf, off); off += 4
        assert buf[off:off + ln].decode() == k; off += ln
        (numel,) = struct.unpack_from("<q", buf, off); off += 8
        assert numel == v.numel()
        assert np.array_equal(np.frombuffer(buf, "<f4", numel, off), v.numpy().ravel()); off += 4 * numel
    assert off == len(buf)


def test_cpp_host_header_compiles_against_the_c_abi():
    """include/yfv2.hpp is header-only over include/yfv2.h: the test driver built by build() must exist and link libyfv2.so."""
    import os
    import subprocess

    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(repo, "tests", "cpp", "yfv2_cpp_test")
    if not os.path.exists(exe):
        import __graft_entry__
        __graft_entry__.build()
    out = subprocess.run(["ldd", exe], capture_output=True, text=True).stdout
    assert "libyfv2.so" in out and "not found" not in out.split("libyfv2.so")[1].splitlines()[0]


def test_ap_per_class_bit_exact_vs_reference_golden():
    """Host arithmetic of evaluation() (utils/utils.py:110-192): float64 results must equal the reference's own
    function bit for bit on the golden sets (tests/golden/make_golden.py ap: ties, unseen classes, degenerate sets)."""
    import os

    import numpy as np

    import yolo_fastestv2_amd as yfv2

    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "golden_ap.npz"))
    for i in range(int(z["n"])):
        got = yfv2.ap_per_class(z["tp%d" % i].astype(np.float64), z["conf%d" % i], z["cls%d" % i], z["labels%d" % i].tolist())
        assert np.array_equal(np.asarray(got, np.float64).view(np.uint64), z["ref%d" % i].view(np.uint64)), (i, got, z["ref%d" % i])
    # compute_ap on a hand-checkable curve: recall 0.5 -> 1.0, precision 1.0 -> 0.5: 0.5*1.0 + 0.5*0.5
    assert yfv2.compute_ap(np.array([0.5, 1.0]), np.array([1.0, 0.5])) == 0.75


def _dryrun(classes, H, W, drop=None, max_batch=4, weights_classes=None):
    import ctypes as C

    import yolo_fastestv2_amd as yfv2
    from yolo_fastestv2_amd import _lib
    from yolo_fastestv2_amd._lib import Config, TensorDesc

    w = yfv2.random_state_dict(1, classes=weights_classes or min(max(classes, 1), 255))
    host = {k: v.float().contiguous() for k, v in w.items() if v.is_floating_point() and k != drop}
    arr = (TensorDesc * len(host))()
    for i, (k, t) in enumerate(host.items()):
        arr[i].name, arr[i].data, arr[i].numel = k.encode(), t.data_ptr(), t.numel()
    cfg = Config()
    cfg.classes, cfg.anchor_num, cfg.height, cfg.width, cfg.max_batch, cfg.device = classes, 3, H, W, max_batch, 0
    ns, nb = C.c_int32(0), C.c_int64(0)
    plan = _lib.make_plan(_lib.plan_from_env())        # (the library reads no environment: the Python layer maps YFV2_FRONT=0 etc. onto yfv2_plan)
    rc = _lib.lib().yfv2_debug_plan_dryrun_ex(C.byref(cfg), C.byref(plan), arr, len(host), C.byref(ns), C.byref(nb))
    return rc, ns.value, nb.value, host


def test_plan_and_weight_packing_on_the_host_for_every_admitted_config():
    """yfv2_debug_plan_dryrun runs the host half of yfv2_create + yfv2_load_weights (configuration check, launch plan,
    BN folding and LDS-image packing) without a device.  Every (classes, size) the configuration check admits must
    plan and pack; what it does not admit must be refused with the CONFIG code, a missing or mis-sized tensor with the
    WEIGHTS code - codes, not crashes."""
    from yolo_fastestv2_amd import _lib

    sizes = [(352, 352), (320, 320), (288, 384), (32, 32), (64, 96), (352, 32), (416, 416), (384, 384), (512, 512), (640, 384), (96, 1024)]
    for classes in (80, 20, 1, 2, 17, 93, 94, 100, 255):
        blobs = set()
        for H, W in sizes:
            rc, steps, blob, _ = _dryrun(classes, H, W)
            assert rc == 0 and steps >= 11 and blob > 400000, (classes, H, W, rc, steps, blob)
            blobs.add(blob)
        assert len(blobs) <= 11     # the packed blob depends on which kernels a size selects, not on the size itself
    # more than 93 classes: the class head no longer fits one chained output conv - it runs as slices of 96 channels
    # (22x22: its tower halves no longer form one step, + objectness head + two class slices: 1 -> 4 + 3; 11x11: its four tower
    # halves no longer form one launch: 1 -> 4 + 3)
    assert _dryrun(100, 352, 352)[1] == _dryrun(80, 352, 352)[1] + 6 + 6
    assert _dryrun(255, 352, 352)[1] == _dryrun(100, 352, 352)[1] + 2                # a third class slice per level
    # the two alternative plans (layer by layer: 77 launches; every pointwise conv on the fp32 MFMA: the stage-3 chain and
    # stage4.0, which exist only as bf16x6 kernels, then run layer by layer) are planned and packed by the same code
    import os
    assert _dryrun(80, 352, 352)[1] == 12      # ten backbone + FPN launches (the stem and stage2.0 are ONE: front2_kernel), one launch for the four 11x11 tower halves, one step for the four 22x22 halves (round 6)
    blob_fused = _dryrun(80, 352, 352)[2]
    os.environ["YFV2_FRONT"] = "0"
    try:
        rc, steps, blob, _ = _dryrun(80, 352, 352)
        assert rc == 0 and steps == 13          # the stem and stage2.0 as two launches (round 4's form)
        assert blob == blob_fused               # front2_kernel reads the two launches' own packed images: the fusion packs nothing new
    finally:
        del os.environ["YFV2_FRONT"]
    os.environ["YFV2_TPAIR"] = "0"
    try:
        assert _dryrun(80, 352, 352)[1] == 15  # the 22x22 tower halves as four launches
    finally:
        del os.environ["YFV2_TPAIR"]
    for var, steps_min in (("YFV2_FUSED", 70), ("YFV2_BF6", 37)):
        os.environ[var] = "0"
        try:
            for classes in (80, 20, 1):
                rc, steps, blob, _ = _dryrun(classes, 352, 352)
                assert rc == 0 and steps >= steps_min, (var, classes, rc, steps)
        finally:
            del os.environ[var]
    ERR_CONFIG, ERR_WEIGHTS = _lib.ERR_CONFIG, _lib.ERR_WEIGHTS
    assert _dryrun(80, 544, 544)[0] == ERR_CONFIG      # 4335 decode rows > the NMS kernel's 4096
    assert _dryrun(80, 640, 640)[0] == ERR_CONFIG
    assert _dryrun(256, 352, 352)[0] == ERR_CONFIG     # the class index travels as one byte
    assert _dryrun(0, 352, 352)[0] == ERR_CONFIG
    assert _dryrun(80, 350, 352)[0] == ERR_CONFIG
    assert _dryrun(80, 352, 352, drop="fpn.conv1x1_2.0.weight")[0] == ERR_WEIGHTS
    assert _dryrun(80, 352, 352, weights_classes=20)[0] == ERR_WEIGHTS     # a 20-class checkpoint into an 80-class handle
    assert _dryrun(20, 352, 352)[2] < _dryrun(80, 352, 352)[2]


def test_bench_quotes_counters_only_from_a_profile_of_the_same_tree(tmp_path, monkeypatch):
    """bench.py's roofline.traffic / kernel_table[].mfma_busy come from profiles/*_traffic.json / *_pmc.json - but only from a
    profile whose source fingerprint (tools/srchash.py) equals the tree's; the lookup matches bench's kernel-name prefixes."""
    import json

    import bench
    h = bench.source_hash()
    assert len(h) == 16 and h == bench.source_hash()
    monkeypatch.setattr(bench, "REPO", str(tmp_path))
    (tmp_path / "profiles").mkdir()
    prof = {"src_hash": h, "kernels": {"void stem_px_kernel<true, false>(StemArgs)": {"total_bytes": 6.0e8, "mfma_busy_pct": 55.5},
                                       "s2px_proj_kernel(S2PxArgs)": {"total_bytes": 2.8e8}, "s2px_main_kernel(S2PxArgs)": {"total_bytes": 3.0e8},
                                       "void tower2_kernel<0, 512, 4, 4, true>(TowerArgs)": {"total_bytes": 8.0e7},
                                       "void tower2_kernel<0, 512, 1, 1, true>(TowerArgs)": {"total_bytes": 2.0e7}}}
    (tmp_path / "profiles" / "r03a_traffic.json").write_text(json.dumps(prof))
    (tmp_path / "profiles" / "r03b_traffic.json").write_text(json.dumps(dict(prof, src_hash="0" * 16)))     # newer name, other tree
    got = bench.newest_profile("_traffic.json", h)
    assert got is not None and got["_file"] == "r03a_traffic.json"
    assert bench.newest_profile("_traffic.json", "f" * 16) is None
    assert bench.profile_lookup(got, "stem_px_kernel", "total_bytes", 1) == 6.0e8
    assert bench.profile_lookup(got, "s2px_proj_kernel + s2px_main_kernel", "total_bytes", 1) == 5.8e8
    assert bench.profile_lookup(got, "tower2_kernel<0, 512, 4, 4,", "total_bytes", 2) == 1.6e8   # two launches per forward
    assert bench.profile_lookup(got, "tower2_kernel<0", "total_bytes", 2) is None                                  # ambiguous prefix: refuse
    assert bench.profile_lookup(got, "stem_px_kernel", "mfma_busy_pct", 1, mean=True) == 55.5
    assert bench.profile_lookup(None, "stem_px_kernel", "total_bytes", 1) is None
    # two instantiations of one template in a counter profile (the fp32 and the uint8 form of the fused front): the one the timed
    # loop ran has by far the most launches; two comparable counts stay ambiguous
    two = {"kernels": {"void front2_kernel<0, false>(FrontArgs)": {"n": 1193, "mfma_busy_pct": 24.6}, "void front2_kernel<0, true>(FrontArgs)": {"n": 4, "mfma_busy_pct": 25.0}}}
    assert bench.profile_lookup(two, "front2_kernel", "mfma_busy_pct", 1, mean=True) == 24.6
    two["kernels"]["void front2_kernel<0, true>(FrontArgs)"]["n"] = 900
    assert bench.profile_lookup(two, "front2_kernel", "mfma_busy_pct", 1, mean=True) is None


def test_committed_counter_profiles_belong_to_this_tree():
    """The evidence rule, applied to the repository itself: profiles/ must hold an HBM-traffic profile and an SQ-counter profile taken on
    THIS source tree (tools/srchash.py fingerprint), or bench.py's roofline.traffic / mfma_busy would print null on the driver's run."""
    import bench
    h = bench.source_hash()
    for suffix in ("_traffic.json", "_pmc.json"):
        got = bench.newest_profile(suffix, h)
        if got is None:
            # Any edit of a kernel source makes the committed profiles stale until someone with an MI355X re-runs the evidence call; that
            # must not turn the CPU suite red.  Stale evidence is reported, and is an ERROR only where the evidence is being
            # produced (tools/gpu_r6.sh exports YFV2_STRICT_EVIDENCE=1 for its final check).
            msg = "no profiles/*%s carries this tree's fingerprint %s: bench.py will print roofline.traffic / mfma_busy as null until tools/gpu_r6.sh is re-run and its summaries are committed" % (suffix, h)
            assert not os.environ.get("YFV2_STRICT_EVIDENCE"), msg
            pytest.skip(msg)
        assert any("front2_kernel" in k for k in got["kernels"]), got["_file"]


def test_data_parallel_training_averages_the_gradient_bucket_world_size_2_gloo(tmp_path):
    """SURVEY.md 8(e) "Training" on CPU: two processes, gloo.  (1) average_gradients_ = one all-reduce + 1/W over a flat
    bucket; (2) Detector.data_parallel(): the train-mode backward averages the whole gradient bucket before autograd sees it
    (a stand-in engine writes rank-dependent gradients through the SAME bind / forward / backward calls the device engine
    gets), so every rank's .grad is the mean over ranks, and the SGD step keeps the replicas identical."""
    script = tmp_path / "w.py"
    script.write_text(textwrap.dedent("""
        import os, sys, torch, torch.distributed as dist
        sys.path.insert(0, %r)
        import yolo_fastestv2_amd as yfv2
        rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
        dist.init_process_group("gloo", rank=rank, world_size=world)
        calls = []
        real = dist.all_reduce
        def counting(*a, **k):
            calls.append(1); return real(*a, **k)
        dist.all_reduce = counting
        flat = torch.arange(10, dtype=torch.float32) * (rank + 1)
        out = yfv2.average_gradients_(flat)
        assert out is flat and torch.equal(flat, torch.arange(10, dtype=torch.float32) * 1.5) and len(calls) == 1

        class StandIn:                                   # the three calls _TrainForward makes on an Engine
            _train_seq = 0
            def train_bind(self, tensors, grads): self.grads = grads
            def train_forward(self, x):
                self._train_seq += 1
                return tuple(torch.zeros(s) for s in self.logit_shapes(x.shape[0]))
            def logit_shapes(self, B): return [(B, 12, 22, 22), (B, 3, 22, 22), (B, 80, 22, 22), (B, 12, 11, 11), (B, 3, 11, 11), (B, 80, 11, 11)]
            def train_backward(self, g6):
                for i, (k, v) in enumerate(self.grads.items()):
                    v += (rank + 1) * (1.0 + 0.001 * i) * float(g6[0].flatten()[0])
        torch.manual_seed(0)                             # same initial weights on both ranks
        model = yfv2.Detector(80, 3, True)
        eng = StandIn()
        model.engine_for = lambda x, sync=True: eng
        model.train()
        model.data_parallel()
        from yolo_fastestv2_amd.model.detector import _TrainForward
        params = [p for _, p in model.named_parameters()]
        opt = torch.optim.SGD(model.parameters(), lr=0.01, momentum=0.949, weight_decay=0.0005)   # yfv2.SGD's kernel needs the device
        before = [p.detach().clone() for p in params]
        outs = _TrainForward.apply(model, torch.zeros(2, 3, 352, 352), *params)
        (2.0 * sum(o.sum() for o in outs)).backward()
        assert len(calls) == 2                           # ONE collective for all 225 gradients
        for i, p in enumerate(params):
            want = 1.5 * (1.0 + 0.001 * i) * 2.0         # mean over ranks of (rank+1) * ...
            assert torch.allclose(p.grad, torch.full_like(p, want), rtol=1e-6), (i, float(p.grad.flatten()[0]), want)
        opt.step()
        mine = torch.cat([p.detach().flatten() for p in params])
        both = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(both, mine)
        assert torch.equal(both[0], both[1])             # replicas stay identical
        assert not torch.equal(mine, torch.cat([b.flatten() for b in before]))
        model.data_parallel(enabled=False)
        outs = _TrainForward.apply(model, torch.zeros(2, 3, 352, 352), *params)
        opt.zero_grad(); sum(o.sum() for o in outs).backward()
        assert len(calls) == 2 and abs(float(params[0].grad.flatten()[0]) - (rank + 1)) < 1e-6
        dist.barrier(); dist.destroy_process_group()
        print("rank", rank, "ok")
    """ % REPO))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29547", WORLD_SIZE="2")
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r)), stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT, text=True) for r in range(2)]
    outs = [p.communicate(timeout=180)[0] for p in procs]
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0 and "rank %d ok" % r in o, o


def test_detect_pipeline_slot_rotation_and_ticket_rules(monkeypatch):
    """DetectPipeline's host logic without a GPU (stand-ins for the Engine and for torch.cuda's streams / events): slots rotate,
    a slot's stream is current while its batch is enqueued, a batch larger than max_batch is refused, a ticket whose slot
    was reused is refused, result() orders the caller's stream behind the ticket's event (or blocks the host)."""
    import contextlib
    import yolo_fastestv2_amd.pipeline as P
    log = []

    class FakeStream:
        def __init__(self, device=None): self.waited = []
        def wait_stream(self, s): self.waited.append(("stream", s))
        def wait_event(self, e): self.waited.append(("event", e))
        def synchronize(self): log.append(("sync", self))

    class FakeEvent:
        def record(self, s): self.stream = s
        def synchronize(self): log.append(("host_wait", self))

    class FakeEngine:
        def __init__(self, device, h, w, classes, anchor_num, anchors=None, max_batch=1, plan=None): self.loaded, self.tripped, self.plan = None, False, plan
        def new_det_buffers(self, B): return (torch.zeros(B, 300, 6), torch.zeros(B, 300, dtype=torch.int32), torch.zeros(B, dtype=torch.int32))
        def load_state_dict(self, sd): self.loaded = sd
        def set_anchors(self, a): self.anchors = a
        def peek_nonfinite(self): log.append(("peek", self)); return self.tripped
        def check_finite(self, what="forward"):
            if self.tripped:
                self.tripped = False                              # (the exact query clears the word)
                raise RuntimeError("range guard: " + what)
        def detect(self, x, conf, iou, out=None, check=True):
            assert check is False                                 # the pipeline looks at the guard itself, BEFORE the rotation advances
            log.append(("detect", self, current[0], tuple(o.shape[0] for o in out)))
            out[2][:] = int(x.sum())
            return out

    current = [None]
    main = FakeStream()

    @contextlib.contextmanager
    def fake_stream_ctx(s):
        prev, current[0] = current[0], s
        try:
            yield
        finally:
            current[0] = prev
    monkeypatch.setattr(P, "Engine", FakeEngine)
    monkeypatch.setattr(P.torch.cuda, "Stream", FakeStream)
    monkeypatch.setattr(P.torch.cuda, "Event", FakeEvent)
    monkeypatch.setattr(P.torch.cuda, "stream", fake_stream_ctx)
    monkeypatch.setattr(P.torch.cuda, "current_stream", lambda device=None: current[0] or main)
    monkeypatch.setattr(torch.Tensor, "record_stream", lambda self, s: None, raising=False)

    pipe = P.DetectPipeline("cpu", 352, 352, 80, 3, anchors=[1.0] * 12, max_batch=4, depth=3)
    pipe.load_state_dict({"w": 1})
    assert all(e.loaded == {"w": 1} for e in pipe.engines) and len(pipe.streams) == 3
    tickets = [pipe.submit(torch.full((2 + (k % 2), 1), float(k + 1)), 0.3, 0.4) for k in range(5)]
    detects = [l for l in log if l[0] == "detect"]
    assert [pipe.engines.index(l[1]) for l in detects] == [0, 1, 2, 0, 1]                  # rotation
    assert [pipe.streams.index(l[2]) for l in detects] == [0, 1, 2, 0, 1]                  # the slot's stream was current
    assert [l[3] for l in detects] == [(2, 2, 2), (3, 3, 3), (2, 2, 2), (3, 3, 3), (2, 2, 2)]   # views of the slot's buffers, B rows
    assert all(s.waited and s.waited[0] == ("stream", main) for s in pipe.streams)         # ordered behind the producer of x
    with pytest.raises(RuntimeError):
        pipe.result(tickets[0])                                                           # slot 0 now holds batch 3
    d, i, c = pipe.result(tickets[3])
    assert ("event", tickets[3].event) in main.waited and int(c[0]) == 4 * 3               # batch 3: three rows of value 4
    pipe.result(tickets[4], host=True)
    assert ("host_wait", tickets[4].event) in log
    with pytest.raises(ValueError):
        pipe.submit(torch.zeros(5, 1), 0.3, 0.4)
    n_wait = sum(len(s.waited) for s in pipe.streams)
    pipe.submit(torch.ones(1, 1), 0.3, 0.4, wait_for_input=False)
    assert sum(len(s.waited) for s in pipe.streams) == n_wait                              # no stream wait for a ready input
    with pipe.slot() as (j, eng, bufs):                                                   # the raw form bench.py uses
        assert current[0] is pipe.streams[j] and eng is pipe.engines[j] and bufs is pipe.buffers[j]
    pipe.synchronize()
    assert sum(1 for l in log if l[0] == "sync") == 3
    # the slot's PREVIOUS batch tripped the range guard.  The next submit on that slot raises, names that batch's ticket,
    # enqueues nothing, and leaves the rotation where it was: the flagged ticket is still the slot's owner (not "reused"), the
    # caller can tell it from the good ones and submit again
    pipe2 = P.DetectPipeline("cpu", 352, 352, 80, 3, anchors=[1.0] * 12, max_batch=4, depth=2, plan={"fp32_matrix": 0})
    assert all(e.plan == {"fp32_matrix": 0} for e in pipe2.engines)
    t0 = pipe2.submit(torch.ones(1, 1), 0.3, 0.4)
    t1 = pipe2.submit(torch.ones(1, 1), 0.3, 0.4)
    pipe2.engines[0].tripped = True                                # slot 0 = ticket t0's batch
    n_det = sum(1 for l in log if l[0] == "detect")
    with pytest.raises(RuntimeError, match="ticket #%d" % t0.serial) as ei:
        pipe2.submit(torch.ones(1, 1), 0.3, 0.4)
    assert ei.value.slot == 0 and ei.value.serial == t0.serial
    assert sum(1 for l in log if l[0] == "detect") == n_det        # nothing was enqueued for the refused batch
    pipe2.result(t0); pipe2.result(t1)                             # both earlier tickets still own their slots
    t2 = pipe2.submit(torch.ones(1, 1), 0.3, 0.4)                  # the word is cleared: the slot takes the next batch
    assert t2.slot == 0 and t2.serial == t1.serial + 1


def test_bench_gpus_2_launches_itself(tmp_path):
    """VERDICT r04 missing 1 / next 2: `python bench.py --gpus N` (what the driver runs) must not need a wrapper - with N > 1 and
    no launcher environment it re-executes itself under torch.distributed.run (127.0.0.1, a free port), every rank joins the
    process group, the steps rotate over the buffer sets with ONE asynchronous packed all-gather each, and rank 0 prints ONE JSON
    line.  Here on CPU: --launcher-selftest swaps the engine for a stand-in and RCCL for gloo (the line is marked as such)."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--launcher-selftest", "--steps", "4", "--warmup", "2", "--blocks", "3"],
                       capture_output=True, text=True, timeout=300, env=env, cwd=str(tmp_path))
    assert r.returncode == 0, (r.stdout[-1000:], r.stderr[-3000:])
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout                                        # rank 0 only
    j = json.loads(lines[0])
    assert j["selftest"] is True and j["n_gpus"] == 2 and j["steps"] == 4 and j["warmup"] == 2 and j["blocks"] == 3
    assert j["launched_by"] == "torch.distributed.run" and j["scaling"] == "weak"
    # ... the driver's largest case: --gpus 8 (eight gloo ranks of the same control flow; every rank's slice checked in the gathered buffer)
    r8 = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "8", "--launcher-selftest", "--steps", "3", "--warmup", "1", "--blocks", "2"],
                        capture_output=True, text=True, timeout=600, env=dict(env, OMP_NUM_THREADS="1"), cwd=str(tmp_path))
    assert r8.returncode == 0, (r8.stdout[-1000:], r8.stderr[-3000:])
    lines8 = [ln for ln in r8.stdout.splitlines() if ln.startswith("{")]
    assert len(lines8) == 1 and json.loads(lines8[0])["n_gpus"] == 8 and json.loads(lines8[0])["launched_by"] == "torch.distributed.run"
    # ... and a plain N = 1 call stays in-process
    r1 = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--launcher-selftest", "--steps", "2", "--blocks", "1"],
                        capture_output=True, text=True, timeout=300, env=env, cwd=str(tmp_path))
    assert r1.returncode == 0 and json.loads([ln for ln in r1.stdout.splitlines() if ln.startswith("{")][-1])["launched_by"] == "plain"


def test_bench_without_a_gpu_fails_loudly(tmp_path):
    """no CPU fallback in the measured path: the real bench refuses to run without an MI355X"""
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--steps", "1"], capture_output=True, text=True, timeout=300, cwd=str(tmp_path))
    assert r.returncode != 0 and "needs an MI355X" in r.stderr


def test_bench_reads_power_and_clock_of_its_own_device_from_sysfs(tmp_path):
    """bench.py's `box.sysfs_under_*_load`: the GPU boxes expose every GPU of their node (and render nodes without sensors);
    the reading must be the one of the device the run uses (matched by PCI bus id through the card's resolved path), in watts and
    MHz, and must fall back to every reporting card when the bus id is unknown."""
    sys.path.insert(0, REPO)
    import bench
    for i, (bus, uw, hz) in enumerate((("0000:26:00.0", 1286000000, 2366000000), ("0000:8e:00.0", 296000000, 2405000000))):
        real = tmp_path / "devices" / "pci0000:00" / bus
        hw = real / "hwmon" / ("hwmon%d" % (3 + i))
        hw.mkdir(parents=True)
        (hw / "power1_input").write_text("%d\n" % uw)
        (hw / "freq1_input").write_text("%d\n" % hz)
        card = tmp_path / "drm" / ("card%d" % (8 * i))
        card.mkdir(parents=True)
        os.symlink(str(real), str(card / "device"))
    empty = tmp_path / "drm" / "card1"                       # a node without sensors
    (empty / "device").mkdir(parents=True)
    root = str(tmp_path / "drm")
    mine = bench.sysfs_power_clock("0000:26:00", root=root)
    assert mine == {"card0": {"power_w": 1286.0, "sclk_mhz": 2366.0, "ours": True}}
    both = bench.sysfs_power_clock(None, root=root)
    assert set(both) == {"card0", "card8"} and both["card8"]["power_w"] == 296.0
    assert bench.sysfs_power_clock("0000:ff:00", root=root).keys() == both.keys()     # unknown bus id: everything that reports
    assert bench.sysfs_power_clock("0000:26:00", root=str(tmp_path / "nothing")) == {}
