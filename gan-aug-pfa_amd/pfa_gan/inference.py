"""Generator-only inference at a fixed batch shape: the hot loop of generate_synthetic_data.py:61-71 as a captured hipGraph
fed by double-buffered pinned-host staging (BASELINE config 5).

    tiles = TilePipeline(gen, batch=4, size=256)
    for index, u8 in tiles(cpu_batches):          # u8: numpy uint8 [B, H, W, 3], the bytes TF.to_pil_image would save
        ...

Per batch the device runs pack -> eval-mode forward -> de-normalise + uint8 (pfa_unpack_u8) as ONE graph replay; the
host->device copy of batch k+1 and the device->host copy of batch k-1 run on a copy stream underneath it, through pinned
buffers, so the loop never blocks on `.cpu()`.  A batch of another shape (the ragged last one) runs eagerly.
"""
import collections

import torch

from . import kernels as K
from ._lib import PfaError


class TilePipeline:
    def __init__(self, gen, batch, size, channels=3):
        if isinstance(size, int):
            size = (size, size)
        p = next(gen.parameters())
        if not p.is_cuda or not gen.fused():
            raise PfaError("TilePipeline needs a GPU generator on the fused engine (model.to('cuda'), default norm layer)")
        if gen.training:
            raise PfaError("TilePipeline is the eval-mode path (generate_synthetic_data.py:55): call gen.eval() first")
        self.gen, self.shape = gen, (batch, channels, size[0], size[1])
        dev = p.device
        eng = gen.engine()
        eng.refresh_shadows(force=False)
        self.eng, self.ctx = eng, eng.new_ctx()
        self.ctx.pin()                           # the captured graphs below hold these buffers' addresses
        self.main = torch.cuda.current_stream(dev)
        self.copy = torch.cuda.Stream(device=dev)
        self.x_pin = [torch.empty(self.shape, dtype=torch.float32).pin_memory() for _ in range(2)]
        self.x_dev = [torch.empty(self.shape, dtype=torch.float32, device=dev) for _ in range(2)]
        oshape = (batch, size[0], size[1], eng.out_nc)
        self.o_dev = [torch.empty(oshape, dtype=torch.uint8, device=dev) for _ in range(2)]
        self.o_pin = [torch.empty(oshape, dtype=torch.uint8).pin_memory() for _ in range(2)]
        self.ev_h2d = [torch.cuda.Event() for _ in range(2)]
        self.ev_done = [torch.cuda.Event() for _ in range(2)]
        self.ev_d2h = [torch.cuda.Event() for _ in range(2)]
        self._used = [False, False]
        for s in range(2):                      # warm-up outside capture: allocates the context buffers and workspaces
            self._forward(s)
        torch.cuda.synchronize(dev)
        self.graphs = []
        for s in range(2):                      # one graph per staging slot (static addresses)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                self._forward(s)
            self.graphs.append(g)
        self.main = torch.cuda.current_stream(dev)

    def _forward(self, s):
        xh = self.ctx.buf("tile.x", (self.shape[0], self.shape[2], self.shape[3], K.pad_ch(self.shape[1], self.eng.dtype)),
                          self.eng.dtype)
        K.pack_nhwc(self.x_dev[s], self.eng.dtype, out=xh)
        fake = self.eng.forward(xh, self.ctx, train=False)
        K.unpack_u8(fake, self.eng.out_nc, 0.5, 0.5, out=self.o_dev[s])

    @torch.no_grad()
    def __call__(self, batches):
        """batches: iterable of CPU float32 NCHW tensors in [-1, 1].  Yields (index, uint8 numpy [B,H,W,C]) in order."""
        pending = collections.deque()
        main, copy = self.main, self.copy
        k = -1
        for k, x in enumerate(batches):
            if tuple(x.shape) != self.shape:                 # ragged batch: drain the pipeline, then the eager path
                while pending:
                    yield self._finish(pending.popleft())
                yield k, self.gen.generate_uint8(x.to(self.x_dev[0].device)).cpu().numpy()
                continue
            s = k % 2
            if self._used[s]:
                self.ev_h2d[s].synchronize()                 # the previous upload from this pinned buffer has left it
            self.x_pin[s].copy_(x)
            with torch.cuda.stream(copy):                    # (stream order: after the previous download of slot s, which
                copy.wait_event(self.ev_done[s]) if self._used[s] else None      # itself waited for that slot's replay)
                self.x_dev[s].copy_(self.x_pin[s], non_blocking=True)
                self.ev_h2d[s].record(copy)
            main.wait_event(self.ev_h2d[s])
            if self._used[s]:
                main.wait_event(self.ev_d2h[s])              # the slot's previous result has been downloaded
            self.graphs[s].replay()
            self.ev_done[s].record(main)
            with torch.cuda.stream(copy):
                copy.wait_event(self.ev_done[s])
                self.o_pin[s].copy_(self.o_dev[s], non_blocking=True)
                self.ev_d2h[s].record(copy)
            self._used[s] = True
            pending.append((k, s))
            if len(pending) == 2:
                yield self._finish(pending.popleft())
        while pending:
            yield self._finish(pending.popleft())

    def _finish(self, item):
        k, s = item
        self.ev_d2h[s].synchronize()
        return k, self.o_pin[s].numpy().copy()
