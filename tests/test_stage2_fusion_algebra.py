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
