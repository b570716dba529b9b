"""The fused Pix2Pix training step: train_gan.train_gan_one_epoch (train_gan.py:46-75) on the HIP engines.

Schedule per batch - the minimal one that reproduces the reference's results:
  D phase  G fwd x1 (BN buffers take the update twice: the reference runs G twice on unchanged weights,
           train_gan.py:56,65) -> D(real) -> D(fake) -> BCE -> D backward x2 -> Adam(D)
  G phase  D(fake) with the updated D -> BCE + 100*L1 -> D input-gradient only -> G backward -> Adam(G)
No torch.cat, no .item() sync inside the step; losses stay on the device until asked for.
"""
import math

import torch

from . import kernels as K
from . import _lib
from ._lib import PfaError
from .engine import FlatParams
from .models import resolve_dtype

PARTS = 4096           # loss partials per pass (GanTrainer._parts); a pass whose kernels would write more keeps the separate loss launch
LAMBDA_L1 = 100.0      # train_gan.py:33
LEARNING_RATE = 1e-4   # train_gan.py:28-29
BETAS = (0.5, 0.999)   # train_gan.py:30,140-141
AUTO_GRAPH_PIXELS = 2 * 256 * 256      # use_graph="auto": hipGraph replay at or below this many pixels per batch


import os as _os
WGRAD_STREAMS = int(_os.environ.get("PFA_WGRAD_STREAMS", "1"))
D_WGRAD_PAIR = _os.environ.get("PFA_D_WGRAD_PAIR", "0") != "0"       # D(fake)'s weight gradients alternate between side2 and side
DREAL_WGRAD_SIDE = _os.environ.get("PFA_DREAL_WGRAD_SIDE", "0") != "0"   # D(real)'s weight gradients on side2 instead of in line
D_EARLY_REDUCE = _os.environ.get("PFA_D_EARLY_REDUCE", "1") != "0"     # data parallel: D's gradients reduced as they become final
BUCKET_UPDATE = _os.environ.get("PFA_BUCKET_UPDATE", "1") != "0"     # Adam + shadow refresh bucket by bucket under the backward
BUCKET_UPDATE_D = _os.environ.get("PFA_BUCKET_UPDATE_D", "0") != "0"   # one GPU: Adam(D) bucket by bucket under the D phase: measured -0.7 % (its range update queues in front of D(fake)'s remaining weight gradients on side2): off
EARLY_G_ZERO = _os.environ.get("PFA_EARLY_G_ZERO", "1") != "0"       # zero the generator's gradient arena under the D phase (A/B switch)
DEFER_DECODER_UPDATE = _os.environ.get("PFA_DEFER_DECODER_UPDATE", "0") != "0"   # measured -2.0 %: off (see GanTrainer._d_update_g_phase)
# dL/dpre from the launch that computes dL/dfake: measured -0.2 % at 256x256 and -0.9 % at 512x512 (the loss-only L1 pass on
# the side stream collides with the HBM-bound first launches of the generator backward): off
ADAM_GRID_BF16 = int(_os.environ.get("PFA_ADAM_GRID", "256"))          # workgroups of the Adam launches in bf16 mode (A/B switch)
L1_FUSED = _os.environ.get("PFA_L1_FUSED", "0") != "0"
# Adam + weight copies in one pass: 25 % less kernel time alone (0.27 against 0.36 ms per step on one stream), but in the
# three-stream step +-0 at 256x256 and -0.8 % at 512x512 (three same-box pairs each): built, tested bit for bit, off
ADAM_SHADOW_FUSED = _os.environ.get("PFA_ADAM_SHADOW_FUSED", "0") != "0"
LAST_BUCKET_MAIN = _os.environ.get("PFA_LAST_BUCKET_MAIN", "1") != "0"   # the last bucket's update on the main stream (A/B switch)


class FusedAdam(torch.optim.Optimizer):
    """optim.Adam(lr, betas) semantics (weight_decay 0, amsgrad off) as one pfa_adam launch per network.

    Takes `model.parameters()` like torch.optim.Adam (train_gan.py:140-141); the parameters must belong to
    one module, whose flat arena (pfa_gan.engine.FlatParams) holds p, and m / v live in matching arenas.
    """

    def __init__(self, params, lr=LEARNING_RATE, betas=BETAS, eps=1e-8):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps))
        if len(self.param_groups) != 1:
            raise PfaError("FusedAdam supports a single parameter group")
        self._flat = None
        self.m = self.v = None
        # workgroups of the Adam launches (0 = the library's 2048).  GanTrainer sets 256 in bf16 mode: one workgroup per CU
        # moves the same bytes per second when the launch runs alone and takes less from the kernels it runs beside - same-box
        # A/B +1.1 ... +1.4 % at B = 16 and 64, +0.7 % at 512 x 512, +0.5 % at batch 1; in fp32 mode -1.0 % (stays 2048)
        self.grid = 0
        self.steps = 0          # host step count: THE step count (eager launches pass it by value)
        self.step_dev = None    # device copy read by captured launches; _dev_steps = the value it holds
        self._dev_steps = 0

    def sync_step_dev(self):
        """Make the device-side counter equal to the host count before a launch that reads it (hipGraph replay).  Eager
        steps (a ragged batch between two replays, step_begin / step_range) advance only the host count."""
        if self._dev_steps != self.steps:
            self.step_dev.fill_(self.steps)
            self._dev_steps = self.steps

    def flat(self):
        if self._flat is None or not self._flat.bound():
            ps = self.param_groups[0]["params"]
            flats = {id(FlatParams.registry.get(id(p))) for p in ps}
            flat = FlatParams.registry.get(id(ps[0]))
            if flat is None or len(flats) != 1 or len(ps) != len(flat.params):
                raise PfaError("FusedAdam needs all parameters of ONE model that is already on the GPU "
                               "(build it with model.parameters() after model.to('cuda'); the arena is created by "
                               "model.engine())")
            if self._flat is not None and self.m is not None and self.m.numel() == flat.total:
                pass                      # model re-bound (e.g. after load): keep the moments
            else:
                self.m = torch.zeros_like(flat.data)
                self.v = torch.zeros_like(flat.data)
                self.step_dev = torch.zeros(1, dtype=torch.int32, device=flat.device)
                self._dev_steps = 0
            self._flat = flat
        return self._flat

    def grad2(self):
        """Second gradient arena (same layout) for a backward pass that runs concurrently on another stream."""
        if getattr(self, "_grad2", None) is None or self._grad2.numel() != self.flat().total:
            self._grad2 = torch.zeros_like(self.flat().grad)
        return self._grad2

    def zero_grad(self, set_to_none=False):
        K.fill_zero(self.flat().grad)
        if getattr(self, "_grad2", None) is not None:
            K.fill_zero(self._grad2)

    @torch.no_grad()
    def step(self, closure=None, grad_scale=1.0, graph_safe=False, use_grad2=False):
        flat = self.flat()
        g = self.param_groups[0]
        g2 = self._grad2 if use_grad2 else None
        if graph_safe:
            self.sync_step_dev()
        self.steps += 1
        flat.epoch += 1          # the arena is written behind torch's version counters
        if graph_safe:      # step count read from device memory so that a captured launch stays valid on replay
            K.adam_tick(self.step_dev)
            self._dev_steps += 1
            K.adam(flat.data, flat.grad, self.m, self.v, g["lr"], g["betas"][0], g["betas"][1], g["eps"], 0,
                   step_dev=self.step_dev, grad_scale=grad_scale, g2=g2, grid=self.grid)
        else:
            K.adam(flat.data, flat.grad, self.m, self.v, g["lr"], g["betas"][0], g["betas"][1], g["eps"], self.steps,
                   grad_scale=grad_scale, g2=g2, grid=self.grid)

    @torch.no_grad()
    def step_begin(self, graph_safe=False):
        """First half of a step taken range by range (step_range): one step count for all ranges.  graph_safe: the count is
        read from device memory (ticked here once), so that a recorded launch list stays valid when it is replayed."""
        self.flat()
        if graph_safe:
            self.sync_step_dev()
        self.steps += 1
        self._flat.epoch += 1
        self._range_dev = bool(graph_safe)
        if graph_safe:
            K.adam_tick(self.step_dev)
            self._dev_steps += 1

    @torch.no_grad()
    def step_range(self, lo, hi, grad_scale=1.0, use_grad2=False):
        """Adam on elements [lo, hi) of the arena (16-byte aligned bounds: parameter starts) with the step count of the
        last step_begin().  The fused trainer updates a bucket as soon as its gradients are final.  use_grad2: the gradient
        is the sum of the two arenas (the discriminator's real | fake passes)."""
        flat = self.flat()
        g = self.param_groups[0]
        g2 = self._grad2[lo:hi] if use_grad2 else None
        if getattr(self, "_range_dev", False):
            K.adam(flat.data[lo:hi], flat.grad[lo:hi], self.m[lo:hi], self.v[lo:hi], g["lr"], g["betas"][0], g["betas"][1],
                   g["eps"], 0, step_dev=self.step_dev, grad_scale=grad_scale, g2=g2, grid=self.grid)
            return
        K.adam(flat.data[lo:hi], flat.grad[lo:hi], self.m[lo:hi], self.v[lo:hi], g["lr"], g["betas"][0], g["betas"][1],
               g["eps"], self.steps, grad_scale=grad_scale, g2=g2, grid=self.grid)

    def state_dict(self):
        self.flat()
        return dict(steps=self.steps, m=self.m.clone(), v=self.v.clone(), param_groups=[
            {k: v for k, v in self.param_groups[0].items() if k != "params"}])

    def load_state_dict(self, sd):
        self.flat()
        self.steps = int(sd["steps"])
        self.m.copy_(sd["m"])
        self.v.copy_(sd["v"])
        self.step_dev.fill_(self.steps)
        self._dev_steps = self.steps
        self.param_groups[0].update(sd["param_groups"][0])


class GanTrainer:
    """Owns the engines, contexts and optimizers of one (generator, discriminator) pair on one GPU."""

    def __init__(self, gen, disc, opt_g=None, opt_d=None, precision="bf16", lambda_l1=LAMBDA_L1, process_group=None,
                 use_graph=False, overlap=True, inputs_resident=False, grad_payload=None, collectives=None):
        """inputs_resident: the caller guarantees that the (packed) batches handed to step_packed() are complete in
        device memory before the call - e.g. a pool of resident batches, or batches produced on a stream the caller
        has synchronised.  The D(real) pass of a step then no longer waits for the previous step's generator update
        (see _d_phase).
        grad_payload: None = the gradient all-reduces carry float32 (the default); "bf16" = every range is cast to
        bfloat16 for the collective and back (half the bytes over xGMI: 89 MB instead of 178 MB per step and rank for the
        default networks; the sum then carries bf16 rounding, like torch DDP's bf16 compression hook).
        collectives: "torch" (default; PFA_COLLECTIVES overrides) = torch.distributed.all_reduce on the group (RCCL's
        own stream, work handles); "c" = the library's communicator (pfa_comm_* / pfa_allreduce: one RCCL communicator
        over the group's ranks, every collective ordered on the stream that issues it).
        use_graph: False = eager launch on three streams from Python; "replay" = the same launches, streams and
        dependencies recorded once per segment as launch lists and re-issued from C (pfa_replay): the host cost of a
        step drops from ~2.5 ms to ~0.2 ms, static shapes and addresses as for graphs; True = the step's three segments
        replayed as captured hipGraphs; "auto" = hipGraph replay on ONE stream when the first batch is tiny (at most
        AUTO_GRAPH_PIXELS = 2 x 256 x 256 pixels per GPU: the reference's own BATCH_SIZE = 1, train_gan.py:26, where the
        step is launch-latency bound and the replay is 26 % faster than eager), eager otherwise."""
        self.gen, self.disc = gen, disc
        self.dtype = resolve_dtype(precision)
        nc_a = gen.input_nc
        nc_b = disc.input_nc - nc_a
        if nc_b != gen.output_nc:
            raise PfaError("discriminator input_nc must equal generator input_nc + output_nc (train_gan.py:139)")
        self.G = gen.engine(self.dtype)
        self.Dn = disc.engine(self.dtype, in_split=(nc_a, nc_b))
        self.opt_g = opt_g if opt_g is not None else FusedAdam(gen.parameters())
        self.opt_d = opt_d if opt_d is not None else FusedAdam(disc.parameters())
        for o in (self.opt_g, self.opt_d):
            if not isinstance(o, FusedAdam):
                raise PfaError("GanTrainer drives pfa_gan.FusedAdam optimizers (same hyper-parameters as optim.Adam)")
        self.opt_g.flat(), self.opt_d.flat()
        self.opt_g.grid = self.opt_d.grid = ADAM_GRID_BF16 if self.dtype == torch.bfloat16 else 0      # (see FusedAdam.grid)
        self.lambda_l1 = float(lambda_l1)
        self.pg = process_group
        if grad_payload not in (None, "bf16"):
            raise PfaError("grad_payload must be None or 'bf16'")
        self.grad_payload = grad_payload if process_group is not None else None
        self._payload = {}
        self.world = torch.distributed.get_world_size(process_group) if process_group is not None else 1
        self._g_reduced = False
        collectives = collectives or _os.environ.get("PFA_COLLECTIVES", "torch")
        if collectives not in ("torch", "c"):
            raise PfaError(f"GanTrainer(collectives={collectives!r}): 'torch' or 'c'")
        # (created here, while every rank is in the constructor: the communicator's set-up is itself a collective)
        self.ccomm = K.Comm.from_process_group(process_group) if process_group is not None and collectives == "c" else None
        self._comm_stream = None             # collectives='c': the one stream every pfa_allreduce of this trainer runs on
        self.skip_collectives = False        # bench.py's comm-off timing pass (never set in training)
        dev = self.G.device
        self.gctx, self.ctx_r, self.ctx_f = self.G.new_ctx(), self.Dn.new_ctx(), self.Dn.new_ctx()
        if overlap or use_graph:
            # the step's side streams read and write these buffers (torch's allocator only knows the stream that allocated
            # them), graphs and launch lists hold their addresses: a replaced buffer is shelved, not freed
            for c in (self.gctx, self.ctx_r, self.ctx_f):
                c.pin()
        self.losses = torch.zeros(2, dtype=torch.float32, device=dev)     # [loss_d, loss_g] of the last step
        self.loss_sums = torch.zeros(2, dtype=torch.float32, device=dev)
        self.nsteps = 0
        self.use_graph = use_graph
        # second HIP stream: parameter-gradient kernels run under the dgrad/BatchNorm chain, D(real) under G forward
        sprio = int(_os.environ.get("PFA_SIDE_PRIO", "0"))       # A/B switch: -1 = the side streams at high priority
        self.side = torch.cuda.Stream(device=dev, priority=sprio) if overlap else None
        self.side2 = torch.cuda.Stream(device=dev, priority=sprio) if overlap else None
        # optional fourth stream: the generator's weight gradients alternate between `side` and this one (PFA_WGRAD_STREAMS=2)
        self.side3 = torch.cuda.Stream(device=dev) if overlap and WGRAD_STREAMS == 2 else None
        self._ev_real_fwd = torch.cuda.Event() if overlap else None
        self._ev_d_free = torch.cuda.Event() if overlap else None     # D engine idle, D weights final for this step
        self._d_free_recorded = False
        self.inputs_resident = bool(inputs_resident)
        self.marks = None
        # Loss sums without float atomics: the loss kernels store one partial per workgroup - D(real) | D(fake) | G's BCE | L1
        # (PARTS each at most) - and one launch at the end of the step adds them in index order: the same bits every run.
        self._parts = torch.zeros(4 * PARTS, dtype=torch.float32, device=dev)
        self._nparts = [0, 0, 0, 0]
        self._graphs = None
        self._listing = False          # launch lists are being warmed up / recorded / replayed (use_graph == "replay")
        self._ev_static = torch.cuda.Event() if overlap else None      # the static input copies of a replayed step are complete
        self._static = None
        self._graph_one_stream = False
        self.plan_shared = _os.environ.get("PFA_PLAN_SHARED") == "1"
        self.G.refresh_shadows(force=True)
        self.Dn.refresh_shadows(force=True)

    def close(self):
        """Release what the trainer holds outside torch: the C-boundary communicator (collectives='c')."""
        if self.ccomm is not None:
            self.ccomm.close()
            self.ccomm = None

    # ----------------------------------------------------------------------------------- pieces of a step
    def _mark(self, name, stream=None):
        """Developer timing (tools/step_marks.py): with self.marks a list, a timing event on `stream` at this point."""
        if self.marks is not None:
            ev = torch.cuda.Event(enable_timing=True)
            ev.record(stream if stream is not None else torch.cuda.current_stream())
            self.marks.append((name, ev))

    def _d_phase(self, xa, xb, graph_safe, ready=None):
        G, Dn = self.G, self.Dn
        parts = self._parts
        garena = self.opt_d.flat().grad
        # one stream: every launch is alone on the GPU (see kernels.set_solo) - except in the tiny-batch hipGraph mode
        # (use_graph="auto"), whose launches do not fill the chip either way: there the hint measured -0.7 %
        K.set_solo(self._solo_default())
        if self.side is None:
            garena2 = self.opt_d.grad2()      # an arena per pass, as on three streams: a second pass ADDING to the first
            self.opt_d.zero_grad()            # one's sums would depend on the order of its float atomics
            fake = G.forward(xa, self.gctx, train=True, bn_repeats=2)
            dl_r = self._d_forward_bce([xa, xb], self.ctx_r, 0, 1.0, 0.5, 0.5)
            dl_f = self._d_forward_bce([xa, fake], self.ctx_f, 1, 0.0, 0.5, 0.5)
            Dn.backward(self.ctx_r, dl_r, garena2, param_grads=True)
            Dn.backward(self.ctx_f, dl_f, garena, param_grads=True)
            K.loss_finish(parts, [(0, self._nparts[0]), (PARTS, PARTS + self._nparts[1])], self.losses[0:1])
            return
        # Three streams.  side: the whole D(real) forward + backward (it needs neither the generator nor D(fake)) into
        # the second gradient arena; main: G forward -> D(fake) forward -> D(fake) dgrad/BatchNorm chain; side2: D(fake)'s
        # parameter gradients.  D(real)'s forward finishes before D(fake)'s starts, so the BatchNorm running buffers are
        # updated in the reference's order (real, then fake: train_gan.py:57,59).
        #
        # D(real) needs the discriminator's weights (final since the Adam(D) of the previous step), its engine state
        # (free once the previous step's G-phase D backward is done: _ev_d_free) and the inputs - nothing of the
        # previous step's generator backward / Adam(G) / shadow refresh / gradient zero-fills.  With inputs_resident
        # the side stream therefore waits for _ev_d_free only, and this step's D(real) runs under the HBM-bound tail of
        # the previous step instead of after it.  Its loss partials have a range of their own (read by this step's loss_finish
        # below, i.e. before the _ev_d_free the NEXT step's D(real) waits for) and its gradient arena is zeroed on the side stream.
        main = torch.cuda.current_stream()
        garena2 = self.opt_d.grad2()
        # data parallel: the discriminator's gradients cross the ranks in three collectives instead of one behind the D phase
        early_reduce = self.pg is not None and not graph_safe and D_EARLY_REDUCE and not D_WGRAD_PAIR and not DREAL_WGRAD_SIDE
        self._d_works = []
        self._d_reduced = False
        # (a captured hipGraph cannot wait for an event of another graph; a launch list can: it re-issues the same waits)
        if (self.inputs_resident or ready is not None) and self._d_free_recorded and (not graph_safe or self._listing):
            K.wait_event(self.side, self._ev_d_free)
            if ready is not None:
                K.wait_event(self.side, ready)
        else:
            K.wait_stream(self.side, main)
        self._mark("step", main)
        K.fill_zero(garena)
        if EARLY_G_ZERO:
            # the generator's gradient arena (167 MB) is zeroed here, on the second side stream under the generator's forward
            # and D(real), instead of in front of the generator phase's discriminator pass, where the main stream runs alone:
            # its readers (the previous step's bucketed Adam) are behind the main stream, its writers (this step's generator
            # backward) behind the main stream's join with side2 after the D phase
            K.wait_stream(self.side2, main)
            with torch.cuda.stream(self.side2):
                self.opt_g.zero_grad()
        with torch.cuda.stream(self.side):
            self._mark("side: D(real) begins")
            K.fill_zero(garena2)
            dl_r = self._d_forward_bce([xa, xb], self.ctx_r, 0, 1.0, 0.5, 0.5)
            K.record_event(self._ev_real_fwd, self.side)
            self._mark("side: D(real) forward done")
            Dn.backward(self.ctx_r, dl_r, garena2, param_grads=True, alt_ws=True, side=self.side2 if DREAL_WGRAD_SIDE else None)
            self._mark("side: D(real) backward done")
            if early_reduce:
                # D(real)'s arena is final here, long before the discriminator's update: its collective runs on the
                # communicator's stream under the rest of the D phase
                self._d_works.append(self._reduce(garena2, ("d", 2), async_op=True))
        fake = G.forward(xa, self.gctx, train=True, bn_repeats=2)
        self._mark("G forward done")
        K.wait_event(main, self._ev_real_fwd)
        dl_f = self._d_forward_bce([xa, fake], self.ctx_f, 1, 0.0, 0.5, 0.5)
        self._mark("D(fake) forward done")
        # (side is free again: D(real) finished under the generator's forward)
        d_lo = Dn.grad_bucket()[0]

        def d_bucket():
            # c4 / c5 and the BatchNorm between them - 76 % of D(fake)'s gradients - are final: their weight gradients on
            # side2, the BatchNorm gradients on the main stream.  The collective is enqueued behind side2 and runs on the
            # communicator's stream under the rest of the backward.
            K.wait_stream(self.side2, main)
            with torch.cuda.stream(self.side2):
                self._d_works.append(self._reduce(garena[d_lo:], ("d", 0), async_op=True))
        bucketed_d = early_reduce
        # One GPU: the discriminator is updated bucket by bucket like the generator - c4 / c5 (76 % of it) on side2 as soon
        # as their gradients are final and their weights have been read for the last time (behind c4's input gradient),
        # under the rest of the backward; only c1 .. c3 are left for the main stream between the D phase and the generator
        # phase's discriminator pass, which waits for this update.
        self._d_updated = False
        update_d = (BUCKET_UPDATE_D and self.pg is None and not D_WGRAD_PAIR and d_lo > 0 and (not graph_safe or self._listing))

        def d_update():
            K.wait_stream(self.side2, main)
            K.wait_stream(self.side2, self.side)          # (D(real)'s arena: complete long ago)
            with torch.cuda.stream(self.side2):
                self.opt_d.step_range(d_lo, self.opt_d.flat().total, grad_scale=1.0, use_grad2=True)
                Dn.refresh_shadows_range(d_lo, self.opt_d.flat().total)
        if update_d:
            self.opt_d.step_begin(graph_safe)
        Dn.backward(self.ctx_f, dl_f, garena, param_grads=True, side=(self.side2, self.side) if D_WGRAD_PAIR else self.side2,
                    bucket_ready=d_bucket if bucketed_d else None, bucket_done=d_update if update_d else None)
        self._mark("D(fake) backward chain done")
        K.wait_stream(main, self.side)
        K.wait_stream(main, self.side2)
        self._mark("D gradients complete")
        if update_d:
            self.opt_d.step_range(0, d_lo, grad_scale=1.0, use_grad2=True)
            Dn.refresh_shadows_range(0, d_lo)
            Dn.shadows_refreshed()
            self._d_updated = True
        if bucketed_d:                       # what is left: c1 .. c3 (and D(real)'s arena, reduced above)
            if d_lo > 0:                     # (n_layers = 1: the bucket already is the whole arena)
                self._d_works.append(self._reduce(garena[:d_lo], ("d", 1), async_op=True))
            self._d_reduced = True
        # loss_d of this step (before _ev_d_free: the next step's D(real) may overwrite its partials from there on)
        K.loss_finish(parts, [(0, self._nparts[0]), (PARTS, PARTS + self._nparts[1])], self.losses[0:1])

    def _solo_default(self):
        # plan_shared (PFA_PLAN_SHARED=1): a one-stream run that keeps the launch plans of the three-stream step - what the
        # bench's roofline pass and the one-stream profiles measure (the kernels of the TIMED configuration, each alone on the GPU)
        return self.side is None and not self._graph_one_stream and not self.plan_shared

    def _d_forward_bce(self, srcs, ctx, slot, target, lscale, gscale):
        """Discriminator forward + BCE-with-logits (loss partials into slot `slot` of self._parts, dlogits into the context):
        one launch fewer where the logit layer's kernel carries the loss (DiscriminatorEngine.forward(bce=)).  -> dlogits"""
        parts = self._parts[slot * PARTS:(slot + 1) * PARTS]
        logits = self.Dn.forward(srcs, ctx, train=True, bce=(target, lscale, gscale, parts))
        dl = ctx.buf("dlogits", logits.shape, logits.dtype)
        n = ctx.v.get("bce_nparts")
        self._nparts[slot] = n if n is not None else K.bce_logits_parts(logits, target, lscale, gscale, parts, dl)
        return dl

    def _d_update_g_phase(self, xa, xb, graph_safe):
        G, Dn = self.G, self.Dn
        if getattr(self, "_d_updated", False):        # already updated bucket by bucket during the D phase
            self._d_updated = False
        elif ADAM_SHADOW_FUSED:
            # Adam and the kernel-side weight copies in one pass over the parameters (pfa_adam_shadow_batch)
            self.opt_d.step_begin(graph_safe)
            Dn.adam_refresh_range(self.opt_d, 0, self.opt_d.flat().total, grad_scale=1.0 / self.world, use_grad2=True)
            Dn.shadows_refreshed()
        else:
            self.opt_d.step(grad_scale=1.0 / self.world, graph_safe=graph_safe, use_grad2=True)
            Dn.refresh_shadows(force=True)
        self._mark("Adam(D) done")
        if self.side is None or not EARLY_G_ZERO:
            self.opt_g.zero_grad()
        fake = self.gctx.v["fake"]
        # (this pass runs alone on the GPU - the generator's backward waits for its result, nothing else is ready - so its
        # launches plan for the whole chip)
        K.set_solo(True)
        try:
            dl_g = self._d_forward_bce([xa, fake], self.ctx_r, 2, 1.0, 1.0, 1.0)
            # the launch that produces dL/dfake (the first layer's input gradient) writes dL/dpre of the generator's
            # output layer directly where it can (kernels.tconv_scatter(l1=)): the L1 + Tanh' pass over the image - 16 us
            # on the critical path - then only has the loss VALUE left to compute, which it does on the second side stream
            dpre = self.gctx.buf("dpre", fake.shape, fake.dtype)
            fused_l1 = L1_FUSED and Dn.l1_fused(self.ctx_r, 1)
            dx = Dn.backward(self.ctx_r, dl_g, None, param_grads=False, need_dx=True, dx_part=1,     # dL/dfake only
                             dx_l1=(fake, xb, G.out_nc, self.lambda_l1, dpre) if fused_l1 else None)
        finally:
            K.set_solo(self._solo_default())
        if self._ev_d_free is not None and (not graph_safe or self._listing):
            K.record_event(self._ev_d_free, torch.cuda.current_stream())    # the next step's D(real) may start from here
            self._d_free_recorded = True
        self._mark("G phase: D forward + input gradient done")
        l1_side = fused_l1 and self.side is not None          # (one stream - also the tiny-batch hipGraph mode: no fork)
        if l1_side:
            K.wait_stream(self.side2, torch.cuda.current_stream())           # (orders the partials behind the previous step's readers)
            with torch.cuda.stream(self.side2):
                self._nparts[3] = K.l1_tanh_bwd_parts(fake, xb, G.out_nc, None, 0, self.lambda_l1, self._parts[3 * PARTS:], None)
        elif fused_l1:
            self._nparts[3] = K.l1_tanh_bwd_parts(fake, xb, G.out_nc, None, 0, self.lambda_l1, self._parts[3 * PARTS:], None)
        else:
            self._nparts[3] = K.l1_tanh_bwd_parts(fake, xb, G.out_nc, dx, 0, self.lambda_l1, self._parts[3 * PARTS:], dpre)
        garena = self.opt_g.flat().grad
        self._works = []
        hook = None
        main = torch.cuda.current_stream()
        buckets = G.grad_buckets()[0]
        # (a launch list re-issues library calls and events - the bucket hooks are just that; a captured hipGraph is kept to
        # the whole-arena update, and so is a list when the buckets' all-reduces - torch calls - would have to be in it)
        bucketed = BUCKET_UPDATE and self.side is not None and self.side2 is not None and (
            not graph_safe or (self._listing and self.pg is None))
        if bucketed:
            # Update bucket by bucket while the backward is still running: a bucket of the gradient arena is final once
            # the side stream (weight gradients) and the main stream (BatchNorm gamma/beta gradients) reach the hook.
            # The second side stream - idle during the generator backward - then takes the bucket: [all-reduce over the
            # ranks ->] Adam on that range -> re-emit the kernel-side copies of its layers.  The decoder's bucket (63 % of
            # the parameters) is updated under the encoder backward; only the last, small bucket is left for the tail.
            self.opt_g.step_begin(graph_safe)
            scale = 1.0 / self.world

            whens = G.grad_buckets()[1]
            pending = []          # buckets whose gradients are final (and, data parallel, on their way) but not yet updated

            def update(ranges):
                """Adam + shadow refresh of gradient ranges on the current stream, contiguous ranges in one launch."""
                ranges = sorted(ranges)
                merged = [list(ranges[0])]
                for lo, hi in ranges[1:]:
                    if lo == merged[-1][1]:
                        merged[-1][1] = hi
                    else:
                        merged.append([lo, hi])
                for lo, hi in merged:
                    if ADAM_SHADOW_FUSED:
                        G.adam_refresh_range(self.opt_g, lo, hi, grad_scale=scale)
                    else:
                        self.opt_g.step_range(lo, hi, grad_scale=scale)
                        G.refresh_shadows_range(lo, hi)

            def hook(i):
                lo, hi = buckets[i]
                if hi <= lo:
                    return
                if i == len(buckets) - 1 and LAST_BUCKET_MAIN:
                    # the last bucket is final when the backward ends: the main stream has nothing left but to wait for
                    # the side streams, so it takes this update itself - one cross-stream hand-off (~20 us each on the
                    # critical path) instead of two (side -> side2 -> main)
                    K.wait_stream(main, self.side)
                    if self.pg is not None:
                        self._reduce(garena[lo:hi], ("g", i), async_op=True).wait()
                    if pending:                                  # (nothing is left pending in the default layout)
                        K.wait_stream(main, self.side2)
                        for _, _, w in pending:
                            if w is not None:
                                w.wait()
                    update([(lo, hi)] + [(a, b) for a, b, _ in pending])
                    del pending[:]
                    return
                K.wait_stream(self.side2, main)
                K.wait_stream(self.side2, self.side)
                with torch.cuda.stream(self.side2):
                    work = None
                    if self.pg is not None:
                        # enqueued behind side2, runs on the communicator's stream; side2 (not the weight-gradient
                        # stream) waits for it - when it updates the range
                        work = self._reduce(garena[lo:hi], ("g", i), async_op=True)
                    pending.append((lo, hi, work))
                    # Adam is HBM-bound (28 B per parameter at ~5.7 TB/s) and so are the chain's launches on the small
                    # maps - the decoder's and encoder's innermost levels stream 8-34 MB of weights each: beside a range
                    # update they took 80-100 us instead of 13-20 (tools/insitu_gantt.py).  The decoder's buckets
                    # therefore wait (their all-reduce does not) until the chain is back on the large maps, i.e. for the
                    # first encoder hook; from there on every bucket is updated as it becomes final.
                    if DEFER_DECODER_UPDATE and whens[i][0] == "dec":
                        return
                    for _, _, w in pending:
                        if w is not None:
                            w.wait()
                    update([(a, b) for a, b, _ in pending])
                    del pending[:]
        elif self.pg is not None and self.side is not None and not graph_safe:
            def hook(i):
                # bucket i is complete once the side stream (weight gradients) and the main stream (BatchNorm
                # gamma/beta gradients) reach this point; the RCCL collective is enqueued behind the side stream
                # and runs on the communicator's own stream while both streams go on with the remaining backward
                lo, hi = buckets[i]
                if hi <= lo:
                    return
                K.wait_stream(self.side, main)
                with torch.cuda.stream(self.side):
                    self._works.append(self._reduce(garena[lo:hi], ("g", i), async_op=True))
        G.backward(self.gctx, dpre, garena,
                   side=(self.side, self.side3) if self.side is not None and self.side3 is not None else self.side,
                   bucket_ready=hook)
        self._mark("G backward chain done")
        if self.side is not None:
            self._mark("side: G weight gradients done", self.side)
        if bucketed:
            self._mark("side2: bucket updates done", self.side2)
        # loss_g of this step (here: the main stream would otherwise only wait for the side streams)
        if l1_side:
            K.wait_stream(torch.cuda.current_stream(), self.side2)      # (the L1 partials were written there, long ago)
        K.loss_finish(self._parts, [(2 * PARTS, 2 * PARTS + self._nparts[2]), (3 * PARTS, 3 * PARTS + self._nparts[3])],
                      self.losses[1:2])
        # (the running loss sums here too, in front of the waits for the side streams: behind them this 5-us launch sat
        # between the last update and the next step's first kernel)
        K.axpy(self.loss_sums, self.losses)
        self._sums_added = True
        if self.side is not None:
            K.wait_stream(torch.cuda.current_stream(), self.side)
        if bucketed:
            K.wait_stream(torch.cuda.current_stream(), self.side2)
            G.shadows_refreshed()
        self._mark("G gradients (and bucketed Adam) complete")
        for w in self._works:
            w.wait()                 # the current stream waits for the collective; no host sync
        self._g_reduced = hook is not None
        self._g_updated = bucketed

    def _g_update(self, graph_safe):
        if getattr(self, "_g_updated", False):       # already updated bucket by bucket during the backward
            self._g_updated = False
        elif ADAM_SHADOW_FUSED:
            self.opt_g.step_begin(graph_safe)
            self.G.adam_refresh_range(self.opt_g, 0, self.opt_g.flat().total, grad_scale=1.0 / self.world)
            self.G.shadows_refreshed()
        else:
            self.opt_g.step(grad_scale=1.0 / self.world, graph_safe=graph_safe)
            self.G.refresh_shadows(force=True)
        if getattr(self, "_sums_added", False):
            self._sums_added = False
        else:
            K.axpy(self.loss_sums, self.losses)
        if self._solo_default():             # the hint is process-wide library state: a step does not leave it set for whatever
            K.set_solo(False)                # launches next (eval forwards, another trainer); phases set it again

    def _reduce(self, t, key, async_op=False):
        """Sum-all-reduce of a float32 gradient range over the ranks on the current stream; returns a handle with .wait()
        (the current stream waits, not the host) when async_op.  With grad_payload == "bf16" the range travels as bfloat16
        through a persistent staging buffer per call site (`key`)."""
        if self.skip_collectives:            # measurement only (bench.py: what the collectives cost the step): ranks diverge
            class _Done:
                def wait(self_inner):
                    pass
            return _Done() if async_op else None
        if self.grad_payload is None and self.ccomm is None:
            return torch.distributed.all_reduce(t, group=self.pg, async_op=async_op)
        buf = None
        if self.grad_payload is not None:
            buf = self._payload.get(key)
            if buf is None or buf.numel() != t.numel():
                buf = self._payload[key] = torch.empty(t.numel(), dtype=torch.bfloat16, device=t.device)
            buf.copy_(t)
        if self.ccomm is not None:
            # The library's communicator.  Every collective of the trainer goes through ONE communication stream, in program
            # order (the same on every rank): RCCL does not allow one communicator's operations to run concurrently from several
            # streams, and the step issues its ranges from three.  The stream waits for the calling stream (the range's
            # producers, the payload copy above); the caller - or whoever calls wait() - waits for the completion event.
            cur = torch.cuda.current_stream()
            if self._comm_stream is None:
                self._comm_stream = torch.cuda.Stream(device=t.device)
            cs = self._comm_stream
            K.wait_stream(cs, cur)
            with torch.cuda.stream(cs):
                self.ccomm.allreduce(t if buf is None else buf)
                if buf is not None:
                    t.copy_(buf)
            done = torch.cuda.Event()
            K.record_event(done, cs)

            class _OnCommStream:
                def wait(self_inner):        # the then-current stream waits; never the host
                    K.wait_event(torch.cuda.current_stream(), done)
            h = _OnCommStream()
            if async_op:
                return h
            h.wait()
            return None
        work = torch.distributed.all_reduce(buf, group=self.pg, async_op=True)

        class _Back:
            def wait(self_inner):
                work.wait()
                t.copy_(buf)
        h = _Back()
        if async_op:
            return h
        h.wait()
        return None

    def _allreduce(self, opt):
        if self.pg is not None:
            if opt is self.opt_g and getattr(self, "_g_reduced", False):
                self._g_reduced = False          # already reduced bucket by bucket during the backward
                return
            if opt is self.opt_d and getattr(self, "_d_reduced", False):
                for w in self._d_works:          # the current stream waits for the collectives; no host sync
                    w.wait()
                self._d_works = []
                self._d_reduced = False
                return
            if opt is self.opt_d:                                # fold the second arena in before the collective
                opt.flat().grad.add_(opt.grad2())
                opt.grad2().zero_()
            self._reduce(opt.flat().grad, "g" if opt is self.opt_g else "d")

    # ----------------------------------------------------------------------------------- public API
    def pack(self, x):
        """NCHW float32 batch (what batch["image1"].to(DEVICE) is, train_gan.py:53-54) -> engine layout."""
        return K.pack_nhwc(x, self.dtype)

    def step_packed(self, xa, xb, ready=None):
        """One G+D step on already packed NHWC inputs.  Returns the device tensor [loss_d, loss_g].
        ready: optional torch.cuda.Event recorded (on whatever stream produced xa / xb) once they are complete; the
        step's streams wait for it, and D(real) may then start before the previous step's tail has finished."""
        if ready is not None:
            torch.cuda.current_stream().wait_event(ready)
        if self.use_graph == "auto":              # decided by the first batch
            self.use_graph = xa.shape[0] * xa.shape[1] * xa.shape[2] <= AUTO_GRAPH_PIXELS
            self._graph_one_stream = bool(self.use_graph)
        if self.use_graph and (self._static is None or self._static[0].shape == xa.shape):
            return self._step_graph(xa, xb)       # (a batch of another shape, e.g. a ragged last one, runs eagerly)
        if not (xa.is_contiguous() and xb.is_contiguous()):      # the kernels take the dense NHWC tensors pack() returns
            xa, xb = xa.contiguous(), xb.contiguous()
        self._d_phase(xa, xb, False, ready if self.side is not None else None)
        self._allreduce(self.opt_d)
        self._d_update_g_phase(xa, xb, False)
        self._allreduce(self.opt_g)
        self._g_update(False)
        self.nsteps += 1
        return self.losses

    def step(self, real_a, real_b):
        xa, xb = self.pack(real_a), self.pack(real_b)
        ready = None
        if self.side is not None:
            # the pack kernels run on the current stream; D(real) starts on the side stream, which with
            # inputs_resident waits for the previous step's events only - so hand it the packs' completion
            ready = torch.cuda.Event()
            ready.record(torch.cuda.current_stream())
        return self.step_packed(xa, xb, ready=ready)

    def _step_graph(self, xa, xb):
        """The same three segments captured once as hipGraphs (static shapes and addresses); the gradient
        all-reduces stay ordinary RCCL calls between them."""
        side = self.side
        if self._graph_one_stream:
            self.side = None                      # capture / replay without the stream forks
        try:
            return self._step_graph_inner(xa, xb)
        finally:
            self.side = side

    def _step_graph_inner(self, xa, xb):
        self._listing = self.use_graph == "replay" and self.side is not None
        ready = self._ev_static if self._listing else None
        if self._graphs is None:
            for c in (self.gctx, self.ctx_r, self.ctx_f):     # captured addresses: replaced buffers must outlive the graphs
                c.pin()
            self._static = (torch.empty_like(xa), torch.empty_like(xb))
            sa, sb = self._static
            sa.copy_(xa)
            sb.copy_(xb)
            # warm-up outside capture: allocates every context buffer and workspace
            for o in (self.opt_g, self.opt_d):    # outside the capture: a captured fill would reset the counter on every replay
                o.sync_step_dev()
            state = self._snapshot()
            if ready is not None:
                ready.record(torch.cuda.current_stream())
            for _ in range(2):
                self._d_phase(sa, sb, True, ready)
                self._d_update_g_phase(sa, sb, True)
                self._g_update(True)
            self._restore(state)
            torch.cuda.synchronize()
            segs = []
            for fn in (lambda: self._d_phase(sa, sb, True, ready), lambda: self._d_update_g_phase(sa, sb, True),
                       lambda: self._g_update(True)):
                segs.append(self._capture(fn))
            self._restore(state)
            self._graphs = segs
            self._capture_stream = torch.cuda.current_stream().cuda_stream
        if self.use_graph == "replay" and torch.cuda.current_stream().cuda_stream != self._capture_stream:
            raise PfaError("GanTrainer(use_graph='replay'): the launch lists were recorded on another current stream")
        sa, sb = self._static
        subst = None
        def takes(x, s):                          # x can stand in for the recorded buffer s as it is
            return x.is_contiguous() and x.shape == s.shape and x.dtype == s.dtype and x.device == s.device
        if self.use_graph == "replay" and takes(xa, sa) and takes(xb, sb):
            # a launch list takes the batch's own buffers: the recorded input addresses are rewritten, nothing is copied
            # (a later replay without subst puts the recorded addresses back: LaunchList.replay)
            subst = {sa.data_ptr(): xa.data_ptr(), sb.data_ptr(): xb.data_ptr()}
        else:
            sa.copy_(xa)
            sb.copy_(xb)
        for o in (self.opt_g, self.opt_d):        # an eager step since the last replay moved only the host count
            o.sync_step_dev()
        if ready is not None:                     # the side stream's D(real) waits for this, not for the previous step's tail
            ready.record(torch.cuda.current_stream())
        if subst is not None:
            self._graphs[0].replay(subst)
            self._allreduce(self.opt_d)
            self._graphs[1].replay(subst)
            self._allreduce(self.opt_g)
            self._graphs[2].replay(subst)
        else:
            self._graphs[0].replay()
            self._allreduce(self.opt_d)
            self._graphs[1].replay()
            self._allreduce(self.opt_g)
            self._graphs[2].replay()
        for o in (self.opt_g, self.opt_d):        # each replay ticks the device counter once
            o.steps += 1
            o._dev_steps += 1
        # the replay wrote parameters and BatchNorm buffers behind torch's version counters: cached eval-mode
        # coefficients (save_samples between epochs) are stale from here
        self.G.bn_epoch += 1
        self.Dn.bn_epoch += 1
        self.nsteps += 1
        return self.losses

    def _capture(self, fn):
        """One segment of the step as an object with .replay(): a captured hipGraph, or - use_graph == "replay" - a launch
        list (pfa_replay): the segment's library calls, event records and stream waits recorded while it runs once, then
        re-issued from C on the same streams (the hipGraph executor serialises the forked branches; this does not)."""
        if self.use_graph == "replay":
            with _lib.Recorder() as rec:
                fn()
            return rec.finish(patch=[t.data_ptr() for t in self._static])
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            fn()
        return g

    def _snapshot(self):
        bufs = [b for m in (self.gen, self.disc) for b in m.buffers()]
        return dict(g=self.opt_g.flat().data.clone(), d=self.opt_d.flat().data.clone(),
                    gm=self.opt_g.m.clone(), gv=self.opt_g.v.clone(), dm=self.opt_d.m.clone(), dv=self.opt_d.v.clone(),
                    gs=(self.opt_g.steps, self.opt_g.step_dev.clone(), self.opt_g._dev_steps),
                    ds=(self.opt_d.steps, self.opt_d.step_dev.clone(), self.opt_d._dev_steps),
                    bufs=[b.clone() for b in bufs], sums=self.loss_sums.clone())

    def _restore(self, st):
        self.opt_g.flat().data.copy_(st["g"]); self.opt_d.flat().data.copy_(st["d"])
        self.opt_g.m.copy_(st["gm"]); self.opt_g.v.copy_(st["gv"])
        self.opt_d.m.copy_(st["dm"]); self.opt_d.v.copy_(st["dv"])
        self.opt_g.steps = st["gs"][0]; self.opt_g.step_dev.copy_(st["gs"][1]); self.opt_g._dev_steps = st["gs"][2]
        self.opt_d.steps = st["ds"][0]; self.opt_d.step_dev.copy_(st["ds"][1]); self.opt_d._dev_steps = st["ds"][2]
        bufs = [b for m in (self.gen, self.disc) for b in m.buffers()]
        for b, s in zip(bufs, st["bufs"]):
            b.copy_(s)
        self.loss_sums.copy_(st["sums"])
        self.G.refresh_shadows(force=True)
        self.Dn.refresh_shadows(force=True)

    def epoch_losses(self, reset=True):
        """(mean loss_d, mean loss_g) since the last reset: ONE device->host sync, not four per step."""
        n = max(self.nsteps, 1)
        ld, lg = (self.loss_sums / n).tolist()
        if reset:
            self.loss_sums.zero_()
            self.nsteps = 0
        return ld, lg


def _trainer_for(gen, disc, opt_g, opt_d, precision=None, process_group=None):
    tr = getattr(gen, "_pfa_trainer", None)
    if tr is None or tr.disc is not disc or tr.opt_g is not opt_g or tr.opt_d is not opt_d or tr.pg is not process_group \
            or not (tr.G.still_bound() and tr.Dn.still_bound()):
        tr = GanTrainer(gen, disc, opt_g, opt_d, precision=precision or gen.precision, process_group=process_group,
                        use_graph="auto")
        object.__setattr__(gen, "_pfa_trainer", tr)
    return tr


# ----------------------------------------------------------------------------------- data parallel helpers
def broadcast_model(module, process_group=None, src=0):
    """Rank `src`'s parameters and buffers become every rank's (start of training / after a resume)."""
    for t in list(module.parameters()) + list(module.buffers()):
        if t.is_contiguous():
            torch.distributed.broadcast(t.data, src, group=process_group)
        else:                                   # channels-last views into the flat arena
            tmp = t.data.contiguous()
            torch.distributed.broadcast(tmp, src, group=process_group)
            t.data.copy_(tmp)


def allreduce_mean_grads(module, process_group=None):
    """Average .grad over the ranks with ONE collective per network (flat float32 buffer), the host-side form of
    what GanTrainer does on its gradient arena.  Works on any backend (gloo on CPU, RCCL on GPU)."""
    ps = [p for p in module.parameters() if p.grad is not None]
    if not ps:
        return
    world = torch.distributed.get_world_size(process_group)
    flat = torch.cat([p.grad.reshape(-1) for p in ps])
    torch.distributed.all_reduce(flat, group=process_group)
    flat.div_(world)
    off = 0
    for p in ps:
        n = p.numel()
        p.grad.copy_(flat[off:off + n].view(p.grad.shape))
        off += n


def load_training_state(ckpt_dir, gen, disc, opt_g, opt_d, device):
    """Resume from the newest generator_epoch_N.pth (+ discriminator, + trainer_epoch_N.pt when present)."""
    import glob
    import os
    import re
    epochs = sorted(int(re.search(r"generator_epoch_(\d+)\.pth$", f).group(1))
                    for f in glob.glob(os.path.join(ckpt_dir, "generator_epoch_*.pth")))
    if not epochs:
        raise FileNotFoundError(f"no generator_epoch_*.pth in {ckpt_dir}")
    e = epochs[-1]
    gen.load_state_dict(torch.load(os.path.join(ckpt_dir, f"generator_epoch_{e}.pth"), map_location=device))
    disc.load_state_dict(torch.load(os.path.join(ckpt_dir, f"discriminator_epoch_{e}.pth"), map_location=device))
    extra = os.path.join(ckpt_dir, f"trainer_epoch_{e}.pt")
    if os.path.exists(extra):
        st = torch.load(extra, map_location=device)
        if isinstance(opt_g, FusedAdam):
            gen.engine(), disc.engine()
        opt_g.load_state_dict(st["opt_g"])
        opt_d.load_state_dict(st["opt_d"])
    return e


def _all_have_batch(have, process_group, device):
    """Data parallel: a batch that collates to None (collate_fn_skip_none, dataset.py:13-18) on ONE rank must be skipped
    on EVERY rank - the others would otherwise enter the gradient all-reduces alone and hang.  Blocking form (one tiny
    MIN all-reduce + host read per iteration); the fused path overlaps the same exchange with the previous step."""
    if process_group is None:
        return have
    flag = torch.tensor([1 if have else 0], dtype=torch.int32, device=device)
    torch.distributed.all_reduce(flag, op=torch.distributed.ReduceOp.MIN, group=process_group)
    return bool(flag.item())


def train_gan_one_epoch(gen, disc, loader, opt_g, opt_d, device=None, lambda_l1=LAMBDA_L1, process_group=None):
    """Same contract as the reference's train_gan_one_epoch (train_gan.py:46-75): iterates `loader` (dict
    batches with "image1"/"image2"), returns (mean loss_d, mean loss_g).

    GPU model + FusedAdam optimizers -> the fused HIP step.  Anything else (CPU model, torch.optim.Adam)
    -> the reference's own sequence of module calls, where a GPU model still runs every forward/backward
    on the HIP engines through autograd.
    """
    gen.train()
    disc.train()
    p = next(gen.parameters())
    device = p.device if device is None else device
    if p.is_cuda and isinstance(opt_g, FusedAdam) and isinstance(opt_d, FusedAdam):
        tr = _trainer_for(gen, disc, opt_g, opt_d, process_group=process_group)
        tr.lambda_l1 = float(lambda_l1)
        tr.epoch_losses(reset=True)
        # Input pipeline: the NEXT batch is copied to the device and packed to NHWC on a stream of its own while the
        # current step runs; the step waits for the batch's event only (PCIe time overlaps compute).
        in_stream = torch.cuda.Stream(device=device)
        main = torch.cuda.current_stream()

        def stage(batch):
            """-> (xa, xb, ready event, agree) - agree() is None on one GPU; under a process group it tells, when the step
            is due, whether EVERY rank has this batch (see _all_have_batch): the MIN all-reduce of the flag is issued
            here, on the input stream, one iteration ahead, so that its result is on the host long before it is read."""
            in_stream.wait_stream(main)         # (allocator safety: buffers released on main are reusable here)
            xa = xb = ev = agree = None
            with torch.cuda.stream(in_stream):
                if batch is not None:
                    a = batch["image1"].to(device, non_blocking=True)
                    b = batch["image2"].to(device, non_blocking=True)
                    xa, xb = tr.pack(a), tr.pack(b)
                    ev = torch.cuda.Event()
                    ev.record(in_stream)
                if process_group is not None:
                    flag = torch.tensor([0 if batch is None else 1], dtype=torch.int32).to(device, non_blocking=True)
                    torch.distributed.all_reduce(flag, op=torch.distributed.ReduceOp.MIN, group=process_group,
                                                 async_op=True).wait()      # the input stream waits, not the host
                    host = torch.empty(1, dtype=torch.int32, pin_memory=True)
                    host.copy_(flag, non_blocking=True)
                    done = torch.cuda.Event()
                    done.record(in_stream)

                    def agree(done=done, host=host, flag=flag):
                        done.synchronize()
                        return bool(host.item())
            for t in (xa, xb):                  # used on the trainer's streams: keep the blocks out of reuse until then
                if t is None:
                    continue
                t.record_stream(main)
                for st in (tr.side, tr.side2):
                    if st is not None:
                        t.record_stream(st)
            return xa, xb, ev, agree

        def run(staged):
            xa, xb, ev, agree = staged
            if agree is not None and not agree():
                return                          # some rank has no batch here: every rank skips the step
            if xa is not None:
                tr.step_packed(xa, xb, ev)

        staged = None
        for batch in loader:
            if batch is None and process_group is None:
                continue
            nxt = stage(batch)
            if staged is not None:
                run(staged)
            staged = nxt
        if staged is not None:
            run(staged)
        return tr.epoch_losses()
    bce = torch.nn.BCEWithLogitsLoss()
    l1 = torch.nn.L1Loss()
    sum_d = sum_g = 0.0
    n = 0
    for batch in loader:
        if not _all_have_batch(batch is not None, process_group, device):
            continue
        real_a, real_b = batch["image1"].to(device), batch["image2"].to(device)
        opt_d.zero_grad()
        fake = gen(real_a).detach()
        pr = disc(torch.cat((real_a, real_b), 1))
        pf = disc(torch.cat((real_a, fake), 1))
        loss_d = (bce(pr, torch.ones_like(pr)) + bce(pf, torch.zeros_like(pf))) * 0.5
        loss_d.backward()
        if process_group is not None:
            allreduce_mean_grads(disc, process_group)
        opt_d.step()
        opt_g.zero_grad()
        fake = gen(real_a)
        pg = disc(torch.cat((real_a, fake), 1))
        loss_g = bce(pg, torch.ones_like(pg)) + l1(fake, real_b) * lambda_l1
        loss_g.backward()
        if process_group is not None:
            allreduce_mean_grads(gen, process_group)      # only G: D's .grad here is the wasted by-product
        opt_g.step()
        sum_d += loss_d.item()
        sum_g += loss_g.item()
        n += 1
    return sum_d / max(n, 1), sum_g / max(n, 1)
