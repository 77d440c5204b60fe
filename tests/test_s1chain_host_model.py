"""CPU check of the seven-blocks-in-one-launch step (block_s1chain_kernel): a numpy model of the KERNEL'S fixed,
lane-uniform data movement - what goes into the LDS tile, what is held one block, what is parked in Z and loaded back -
driven only by what the HOST packed for that launch (the seven block images with their PS / PL tables, yfv2_debug_plan_image)
and the channel order the plan reports for the output (yfv2_debug_plan_c2_label), against the oracle's seven stride-1
blocks.  This pins the planner (PlanBuilder::s1chain_block: pw1 column / pw2 row permutations, park positions, load-back
tables, output labels) and the index algebra it shares with the kernel; the HIP code itself needs the GPU tests."""
import ctypes as C

import numpy as np
import pytest
import torch

import yolo_fastestv2_amd as yfv2
from oracle import yfv2_oracle as oracle
from yolo_fastestv2_amd import _lib
from yolo_fastestv2_amd._lib import Config, TensorDesc

KC, C2, NB = 3, 48, 7
W_FL, DW_FL, CST_FL, TBL_FL = KC * KC * 256, 9 * KC * 16, 6 * KC * 16, 64
IMG_FL = 2 * W_FL + DW_FL + CST_FL + TBL_FL
BF6 = False          # which image form the helpers below decode; set by _set_form()


def _set_form(bf6):
    """fp32 fragments [3][3][64][4] (block_s1chain_kernel, YFV2_S1CHAIN_BF6=0) or the pre-split form of the default
    block_s1chain6_kernel: per filter [mt (3)][six 16-byte operands][64][4]"""
    global W_FL, IMG_FL, BF6
    BF6 = bf6
    W_FL = 3 * 3 * 256 if bf6 else KC * KC * 256
    IMG_FL = 2 * W_FL + DW_FL + CST_FL + TBL_FL


def _h(u):
    return u.astype(np.uint16).view(np.float16).astype(np.float64)


def _decode6(fr):
    """[mt][{w1, w1} of the chunk pair | {w2, w2} of the pair | {w1, w2} of chunk 2][lane][4 dwords] -> (48, 48) float64 =
    the filter x 2^sw as the sum of its two fp16 terms; every dword = two fp16 (low half first)."""
    u = fr.view(np.uint32).reshape(3, 3, 64, 4)
    m = np.zeros((2, C2, C2))
    for mt in range(3):
        for l in range(64):
            r, g = 16 * mt + (l & 15), l >> 4
            for term in range(2):
                for d in range(4):
                    c = 16 * (d >> 1) + 4 * g + 2 * (d & 1)
                    m[term, r, c], m[term, r, c + 1] = _h(u[mt, term, l, d] & 0xFFFF), _h(u[mt, term, l, d] >> 16)
            for d in range(4):
                c = 32 + 4 * g + 2 * (d & 1)
                m[d >> 1, r, c], m[d >> 1, r, c + 1] = _h(u[mt, 2, l, d] & 0xFFFF), _h(u[mt, 2, l, d] >> 16)
    assert 2.0 ** 13 < np.abs(m[0] + m[1]).max() <= 2.0 ** 14
    return m[0] + m[1]


def _descs(w):
    host = {k: v.float().contiguous() for k, v in w.items() if v.is_floating_point()}
    arr = (TensorDesc * len(host))()
    for i, (k, t) in enumerate(host.items()):
        arr[i].name, arr[i].data, arr[i].numel = k.encode(), t.data_ptr(), t.numel()
    cfg = Config()
    cfg.classes, cfg.anchor_num, cfg.height, cfg.width, cfg.max_batch, cfg.device = 80, 3, 352, 352, 1, 0
    return host, arr, cfg


def _plan(w):
    host, arr, cfg = _descs(w)
    L = _lib.lib()
    ns, nb = C.c_int32(0), C.c_int64(0)
    assert L.yfv2_debug_plan_dryrun(C.byref(cfg), arr, len(host), C.byref(ns), C.byref(nb)) == 0
    name = C.create_string_buffer(256)
    buf = np.zeros(NB * IMG_FL, np.float32)
    im = None
    for st in range(ns.value):
        n = L.yfv2_debug_plan_image(C.byref(cfg), arr, len(host), st, name, 256, buf.ctypes.data_as(C.c_void_p), buf.size)
        if n > 0 and "chain of %d fused s1 blocks" % NB in name.value.decode():
            assert n >= NB * IMG_FL
            im = buf.copy()
            break
    lab = (C.c_int32 * 96)()
    rc = L.yfv2_debug_plan_c2_label(C.byref(cfg), arr, len(host), lab)
    return im, rc, np.asarray(list(lab))


def _frag_matrix(fr):
    m = np.zeros((C2, C2), np.float32)
    fr = fr.reshape(KC, KC, 64, 4)
    for mt in range(KC):
        for s in range(KC):
            for l in range(64):
                m[16 * mt + (l & 15), 16 * s + 4 * (l >> 4):16 * s + 4 * (l >> 4) + 4] = fr[mt, s, l]
    return m


def _branch(tile_phys, im):
    """tile_phys: (H, W, 48) branch input in PHYSICAL tile order -> (H, W, 3 mt, 4 g, 4 e) accumulators after pw2 + BN + ReLU"""
    dec = _decode6 if BF6 else _frag_matrix
    w1, w2 = dec(im[:W_FL]), dec(im[W_FL:2 * W_FL])
    wd = im[2 * W_FL:2 * W_FL + DW_FL].reshape(9, C2)
    cs = im[2 * W_FL + DW_FL:2 * W_FL + DW_FL + CST_FL].reshape(6, C2)
    H, W, _ = tile_phys.shape
    up = 16.0 if BF6 else 1.0                         # fp16x3: activations x 2^4, filters x 2^sw, both undone inside the BN scale
    y = np.maximum((tile_phys * up) @ w1.T * cs[0] + cs[1], 0.0).astype(np.float32)
    pad = np.zeros((H + 2, W + 2, C2), np.float32)
    pad[1:-1, 1:-1] = y
    d = np.zeros((H, W, C2), np.float32)
    for k in range(9):
        d += pad[k // 3:k // 3 + H, k % 3:k % 3 + W] * wd[k]
    d = d * cs[2] + cs[3]
    return np.maximum((d * up) @ w2.T * cs[4] + cs[5], 0.0).astype(np.float32).reshape(H, W, 3, 4, 4)


def _tables(im):
    t = im[2 * W_FL + DW_FL + CST_FL:IMG_FL].view(np.int32)
    return t[:12].reshape(3, 4), t[12:36].reshape(6, 4)       # PS[mt][g] (park positions of elements 2), XS[c][g] (block 0: of X[16 c + 4 g])


def _park(z, ps, g, mt, v):
    """elements 2: three dwords at PS[0..2][g]; the planner keeps the three of lane groups 0..2 adjacent (one consumer, one cache line)"""
    pos = ps[0, g] + mt if g < 3 else ps[mt, 3]
    assert pos == ps[mt, g], "park table: lane group %d's run is not consecutive" % g
    z[..., pos] = v


def _kernel_model(x, images):
    """x: (H, W, 96) -> Z: (H, W, 96) in the kernel's physical order, following block_s1chain_kernel lane group by lane group"""
    H, W, _ = x.shape
    z = np.full((H, W, 96), np.nan, np.float32)
    xq = x.reshape(H, W, 6, 4, 4)                      # [quad c][lane group g][element]
    tile = np.zeros((H, W, C2), np.float32)            # physical position 16 j + 4 g + e  (plane 4 j + g, element e)
    for g in range(4):
        for j in range(3):
            tile[..., 16 * j + 4 * g + 0] = xq[:, :, 2 * j, g, 1]
            tile[..., 16 * j + 4 * g + 1] = xq[:, :, 2 * j, g, 3]
            tile[..., 16 * j + 4 * g + 2] = xq[:, :, 2 * j + 1, g, 1]
            tile[..., 16 * j + 4 * g + 3] = xq[:, :, 2 * j + 1, g, 3]
    hold2 = xq[:, :, :, :, 2].copy()                   # [c][g]
    ps, xs = _tables(images[0])
    for c in range(6):
        for g in range(4):
            pos = xs[c, 0] if g == 0 else xs[3 * (c // 3), g] + c % 3      # lane groups 1..3: two 12-byte runs
            assert pos == xs[c, g], "X park table: a run of lane group %d is not consecutive" % g
            assert np.isnan(z[..., pos]).all(), "two X values parked at one Z position"
            z[..., pos] = xq[:, :, c, g, 0]

    def fresh_quads(bo, g):
        return ([None, None, bo[:, :, 0, g, 1], bo[:, :, 0, g, 3]], [bo[:, :, 1, g, 1], bo[:, :, 1, g, 3], bo[:, :, 2, g, 1], bo[:, :, 2, g, 3]])

    def put(tile_n, g, q0, q1, q2):
        for j, q in enumerate((q0, q1, q2)):
            for e in range(4):
                tile_n[..., 16 * j + 4 * g + e] = q[e]

    # block 0 and its exchange (X's held elements 2)
    im = images[0]
    bo = _branch(tile, im)
    ps, _ = _tables(im)
    tile_n = np.zeros_like(tile)
    hd = np.zeros((H, W, 3, 4), np.float32)
    for g in range(4):
        q1, q2 = fresh_quads(bo, g)
        q1[0], q1[1] = hold2[:, :, 4, g], hold2[:, :, 5, g]
        put(tile_n, g, [hold2[:, :, c, g] for c in range(4)], q1, q2)
        for mt in range(3):
            _park(z, ps, g, mt, bo[:, :, mt, g, 2])
            hd[:, :, mt, g] = bo[:, :, mt, g, 0]
    tile = tile_n
    for kb in range(1, NB):
        im = images[kb]
        ps, _ = _tables(im)
        more = kb + 1 < NB
        plv = np.zeros((H, W, 3, 4), np.float32)
        if more:                                       # the next block's twelve parked inputs: group kb - 1 of Z, three per lane group
            for i in range(3):
                for g in range(4):
                    plv[:, :, i, g] = z[..., 12 * (kb - 1) + 3 * g + i]
            assert not np.isnan(plv).any(), "block %d loads back a Z position nobody has written yet" % (kb + 1)
        bo = _branch(tile, im)
        if more:
            tile_n = np.zeros_like(tile)
            for g in range(4):
                q1, q2 = fresh_quads(bo, g)
                q1[0], q1[1] = plv[:, :, 1, g], plv[:, :, 2, g]
                put(tile_n, g, [hd[:, :, 0, g], hd[:, :, 1, g], hd[:, :, 2, g], plv[:, :, 0, g]], q1, q2)
            for g in range(4):
                for mt in range(3):
                    _park(z, ps, g, mt, bo[:, :, mt, g, 2])
            hd = bo[:, :, :, :, 0].copy()
            tile = tile_n
    for g in range(4):
        for mt in range(3):
            for e in range(4):
                z[..., 16 * mt + 4 * g + e] = bo[:, :, mt, g, e]
            z[..., 48 + 3 * g + mt] = hd[:, :, mt, g]
    assert not np.isnan(z).any(), "a Z position was never stored"
    return z


def test_chain_host_packing_and_channel_bookkeeping():
    _set_form(True)      # block_s1chain6_kernel's pre-split image (the fp32-fragment form went with round 2's fp32-MFMA chain)
    w = yfv2.random_state_dict(5)
    im, rc, lab = _plan(w)
    if im is None:
        pytest.skip("this build's plan has no chain launch")
    assert rc == 1 and sorted(lab.tolist()) == list(range(96)), "the plan's C2 labels are not a permutation"
    torch.manual_seed(0)
    x = torch.randn(1, 96, 22, 22)
    ref = x
    for k in range(1, 8):
        ref = oracle._shuffle_block(w, "backbone.stage3.%d" % k, ref, 1)
    z = _kernel_model(x[0].permute(1, 2, 0).contiguous().numpy(), [im[k * IMG_FL:(k + 1) * IMG_FL] for k in range(NB)])
    got = np.empty_like(z)
    got[..., lab] = z                                   # physical position k holds logical channel lab[k]
    ref = ref[0].permute(1, 2, 0).numpy()
    err = np.abs(got - ref).max()
    assert err <= 1e-4 * max(1.0, np.abs(ref).max()), "chain dataflow model vs oracle: max abs err %g" % err
