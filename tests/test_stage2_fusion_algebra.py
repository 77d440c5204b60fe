"""Channel algebra of stage 2's three stride-1 ShuffleV2 blocks (model/backbone/shufflenetv2.py:48-51,57-63: channel_shuffle
sends the even positions past the block and the odd positions through branch_main; the output is cat(passthrough, main)) -
the bookkeeping a fused stage2.1-3 launch would be built on: which of the 48 input channels
each block reads, which results never leave the registers, what reaches memory.  Symbolic (labels, no arithmetic) and
checked against the oracle's own block function on random data for the routing."""
import numpy as np
import torch

import yolo_fastestv2_amd as yfv2
from oracle import yfv2_oracle as oracle


def _block(x, k):
    """one stride-1 block on a list of labels: even positions pass, odd positions feed main -> 24 new labels (k, j)"""
    return x[0::2] + [("y%d" % k, j) for j in range(len(x) // 2)], x[1::2]


def test_three_stride1_blocks_read_42_channels_and_write_42():
    x = [("a", c) for c in range(48)]
    reads, outs = [], x
    for k in (1, 2, 3):
        outs, main_in = _block(outs, k)
        reads.append(main_in)
    from_memory = [[c for (s, c) in r if s == "a"] for r in reads]
    assert from_memory[0] == list(range(1, 48, 2))                        # block 1: the 24 odd channels
    assert from_memory[1] == list(range(2, 48, 4))                        # block 2: 12 channels 2, 6, .., 46
    assert from_memory[2] == list(range(4, 48, 8))                        # block 3: 6 channels 4, 12, .., 44
    assert sum(len(f) for f in from_memory) == 42
    in_registers = [[(s, c) for (s, c) in r if s != "a"] for r in reads]
    assert in_registers[1] == [("y1", j) for j in range(1, 24, 2)]        # block 2 takes y1's odd channels from the block before it
    assert in_registers[2] == [("y1", j) for j in range(2, 24, 4)] + [("y2", j) for j in range(1, 24, 2)]   # block 3: 6 of y1 (two rows older), 12 of y2
    final = outs
    assert final[:6] == [("a", c) for c in range(0, 48, 8)]              # six input channels are never read or written by any block
    assert final[6:12] == [("y1", j) for j in range(0, 24, 4)]
    assert final[12:24] == [("y2", j) for j in range(0, 24, 2)]
    assert final[24:] == [("y3", j) for j in range(24)]
    written = [lab for lab in final if lab[0] != "a"]
    assert len(written) == 42                                             # 6 + 12 + 24 results reach memory; 12 + 6 + 12 never do


def test_routing_matches_the_oracle_blocks():
    """the same routing with the oracle's arithmetic: stage 2's output channel q equals what the symbolic model says it is"""
    w = yfv2.random_state_dict(3)
    torch.manual_seed(0)
    x = torch.rand(1, 3, 64, 64)
    with torch.no_grad():
        y = oracle.forward_stages(w, x)["stem"]
        acts = []
        for i in range(4):                                                 # stage2.0 (stride 2), then the three stride-1 blocks
            y = oracle._shuffle_block(w, "backbone.stage2.%d" % i, y, 2 if i == 0 else 1)
            acts.append(y[0].numpy())
    assert np.array_equal(acts[3], oracle.forward_stages(w, x)["stage2"][0].numpy())
    a0, out = acts[0], acts[3]
    assert np.array_equal(out[:6], a0[0::8])                              # untouched channels, bit for bit
    y1, y2 = acts[1][24:], acts[2][24:]
    assert np.array_equal(out[6:12], y1[0::4]) and np.array_equal(out[12:24], y2[0::2])


def test_lane_group_assignment_needs_no_movement_between_lane_groups():
    """The matrix-core layout of the strip kernels (yfv2_stage2h.hip): a 24-channel pointwise conv leaves lane group g of a pixel
    with accumulators (tile 0: 4 per group g = 0..3; tile 1: 4 per group g = 0, 1) and wants its 24 inputs as K slots 8g..8g+7
    of groups 0..2.  Which OUTPUT channel lands in which accumulator is the host's choice (filter rows), which INPUT channel
    sits in which K slot too (filter columns).  A fused stage2.1-3 is free of cross-lane traffic if every intermediate channel
    can be produced in the lane group that later consumes it - constructed and checked here."""
    cap = {(0, g): 4 for g in range(4)}
    cap.update({(1, 0): 4, (1, 1): 4})
    # y1: odd channels feed block 2, channels 2 mod 4 feed block 3 (two rows later), channels 0 mod 4 are final
    y1 = {(0, 0): [1, 3, 5, 7], (0, 1): [9, 11, 13, 15], (0, 2): [17, 19, 21, 23],          # -> block 2, K slots of groups 0, 1, 2
          (1, 0): [2, 6, 10, 14], (1, 1): [18, 22, 0, 4],                                   # -> block 3 (six of them), two finals
          (0, 3): [8, 12, 16, 20]}                                                           # finals
    # y2: odd channels feed block 3, even channels are final
    y2 = {(0, 0): [1, 3, 5, 7], (0, 1): [9, 11, 13, 15], (0, 2): [17, 19, 21, 23],
          (1, 0): [0, 2, 4, 6], (1, 1): [8, 10, 12, 14], (0, 3): [16, 18, 20, 22]}
    for y in (y1, y2):
        assert sorted(c for v in y.values() for c in v) == list(range(24))
        assert all(len(v) <= cap[k] for k, v in y.items())
    group_of = lambda y, c: next(g for (t, g), v in y.items() if c in v)                  # noqa: E731
    # block 2: 12 register inputs (y1 odd) + 12 from memory; block 3: 12 (y2 odd) + 6 (y1 2 mod 4) + 6 from memory
    k2 = {g: [("y1", c) for c in range(1, 24, 2) if group_of(y1, c) == g] for g in range(3)}
    k3 = {g: [("y2", c) for c in range(1, 24, 2) if group_of(y2, c) == g] + [("y1", c) for c in range(2, 24, 4) if group_of(y1, c) == g] for g in range(3)}
    assert sum(len(v) for v in k2.values()) == 12 and sum(len(v) for v in k3.values()) == 18     # nothing needed sits in lane group 3
    mem2 = {g: 8 - len(k2[g]) for g in range(3)}
    mem3 = {g: 8 - len(k3[g]) for g in range(3)}
    assert all(v >= 0 for v in mem2.values()) and all(v >= 0 for v in mem3.values())
    assert sum(mem2.values()) == 12 and sum(mem3.values()) == 6                                    # = the channels the algebra test says come from memory
    assert all(v % 2 == 0 for v in mem2.values()) and all(v % 2 == 0 for v in mem3.values())       # whole 8-byte pairs per lane
    # stores: finals of y1 (0 mod 4) and y2 (even) pair up inside one lane's accumulators of one tile
    for y, final in ((y1, set(range(0, 24, 4))), (y2, set(range(0, 24, 2)))):
        for (t, g), v in y.items():
            assert len([c for c in v if c in final]) % 2 == 0, (t, g, v)
