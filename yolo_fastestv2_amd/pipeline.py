"""Independent batches in flight on several handles.

Every launch of this path has an under-filled tail - stages 3 / 4 and the towers run one workgroup per image, the strip
kernels end on a last round of waves, decode + NMS is one workgroup per image with half of them idle towards the end - and
a handle runs its launches strictly one after the other on one stream.  Batches are independent, so a
caller with a QUEUE of batches keeps `depth` of them in flight: `depth` handles (same weights, own workspaces and
detection buffers) on `depth` HIP streams, consecutive batches rotating over them.  Three in flight finish 11-12 % sooner
per batch than back to back on one MI355X (tools/pipeline_probe.py); more than three gain nothing.

    pipe = DetectPipeline(device, 352, 352, 80, 3, anchors=cfg["anchors"], max_batch=256)
    pipe.load_state_dict(state_dict)
    tickets = [pipe.submit(x, 0.3, 0.4) for x in batches]          # returns at once; results are device tensors
    for t in tickets[-pipe.depth:]:
        dets, idx, cnt = pipe.result(t)                            # orders the CURRENT stream behind that batch

A ticket's buffers belong to its slot: they are overwritten by the `depth`-th submit after it.  A consumer that reads them
on its own stream calls `pipe.release(ticket)` after enqueueing its last reader; the slot's next submit then waits for that
point (without it, only a `result()` issued AFTER the reuse is caught - by the serial check).
"""
import contextlib

import torch

from .engine import Engine


class Ticket:
    __slots__ = ("slot", "event", "out", "serial")

    def __init__(self, slot, event, out, serial):
        self.slot, self.event, self.out, self.serial = slot, event, out, serial


class DetectPipeline:
    def __init__(self, device, height, width, classes, anchor_num, anchors=None, max_batch=1, depth=3, plan=None):
        if depth < 1:
            raise ValueError("depth must be >= 1")
        self.device = torch.device(device)
        self.depth = int(depth)
        self.engines = [Engine(self.device, height, width, classes, anchor_num, anchors=anchors, max_batch=max_batch, plan=plan) for _ in range(self.depth)]
        self.streams = [torch.cuda.Stream(device=self.device) for _ in range(self.depth)]
        self.buffers = [e.new_det_buffers(max_batch) for e in self.engines]
        self.max_batch = int(max_batch)
        self._serial = 0
        self._last = [None] * self.depth          # serial of the ticket that owns each slot's buffers
        self._readers = [[] for _ in range(self.depth)]   # events recorded on the consumer streams result() ordered behind a slot

    def load_state_dict(self, state_dict):
        for e in self.engines:
            e.load_state_dict(state_dict)

    def set_anchors(self, anchors):
        for e in self.engines:
            e.set_anchors(anchors)

    @contextlib.contextmanager
    def slot(self):
        """Next slot in rotation with its stream current: `with pipe.slot() as (j, engine, buffers): engine.detect(x, .., out=buffers)`.
        For callers that enqueue more than a detect per batch (bench.py adds the RCCL gather)."""
        j = self._serial % self.depth
        self._serial += 1
        self._last[j] = self._serial
        with torch.cuda.stream(self.streams[j]):
            yield j, self.engines[j], self.buffers[j]

    def submit(self, x, conf_thres, iou_thres, wait_for_input=True):
        """Enqueue forward + decode + NMS of one batch (fp32 (B,3,H,W) or uint8 (B,H,W,3) on the device, B <= max_batch) on the next
        slot.  wait_for_input: order the slot's stream behind the CURRENT stream first (x was produced there); pass False for an
        input that is already complete - the wait costs a few microseconds of bubble per batch."""
        B = int(x.shape[0])
        if B > self.max_batch:
            raise ValueError("batch %d exceeds max_batch %d" % (B, self.max_batch))
        cur = torch.cuda.current_stream(self.device)
        # The slot's previous batch may have tripped the range guard.  Looked at BEFORE the rotation advances: the error then names that
        # batch's ticket, which stays valid (its buffers are not overwritten - nothing is enqueued for the new batch), and the caller
        # can submit again.  (Inside the slot the same look would clear the word with the rotation already advanced: the flagged
        # ticket would read "reused by a later submit" and could no longer be told from a good one.)
        jn = self._serial % self.depth
        if self.engines[jn].peek_nonfinite():
            try:
                with torch.cuda.stream(self.streams[jn]):
                    self.engines[jn].check_finite("DetectPipeline.submit: the batch of ticket #%s (slot %d), the last one run on this slot" % (self._last[jn], jn))
            except Exception as e:
                e.slot, e.serial = jn, self._last[jn]
                raise
        with self.slot() as (j, eng, (dets, idx, cnt)):
            # write-after-read: streams that result() ordered behind this slot's previous batch may still be reading its
            # detection buffers - the slot's stream waits for the marks they left before it overwrites them
            for ev in self._readers[j]:
                self.streams[j].wait_event(ev)
            self._readers[j] = []
            if wait_for_input:
                self.streams[j].wait_stream(cur)
                x.record_stream(self.streams[j])
            out = eng.detect(x, conf_thres, iou_thres, out=(dets[:B], idx[:B], cnt[:B]), check=False)
            ev = torch.cuda.Event()
            ev.record(self.streams[j])
        return Ticket(j, ev, out, self._serial)

    def result(self, ticket, host=False):
        """(dets, idx, cnt) of a ticket.  Orders the current stream behind the batch (host=True: blocks the host instead)."""
        if self._last[ticket.slot] != ticket.serial:
            raise RuntimeError("this ticket's buffers were reused by a later submit (a slot is overwritten %d submits later)" % self.depth)
        if host:
            ticket.event.synchronize()
            if self.engines[ticket.slot].peek_nonfinite():       # the batch is complete: what the range guard holds is exact
                self.engines[ticket.slot].check_finite("DetectPipeline.result")
        else:
            torch.cuda.current_stream(self.device).wait_event(ticket.event)
        return ticket.out

    def release(self, ticket):
        """Mark the point on the CURRENT stream after which the ticket's buffers are no longer read there: the slot's next
        submit waits for it.  Call after the last consumer kernel of `result(ticket)` was enqueued (a host=True consumer that
        copied the tensors needs no mark)."""
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(self.device))
        self._readers[ticket.slot].append(ev)

    def synchronize(self):
        for s in self.streams:
            s.synchronize()
