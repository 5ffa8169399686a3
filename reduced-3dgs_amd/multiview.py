"""View-parallel training step support: one process per MI355X, one camera per rank per step.

The reference is single-GPU (utils/general_utils.py:133 pins cuda:0; train.py renders one view per
iteration).  Each view's render pass is independent given replicated Gaussians, so the 8 GPUs of a node
render 8 different cameras; before the optimizer / densify / prune step the ranks exchange exactly what the
single-GPU loop would have accumulated over those views:

  * SUM  of the parameter gradients of the rasterizer inputs (means3D 3, SH 3*M, opacity 1, scales 3,
         rotations 4 floats per Gaussian)           -> what Adam consumes (train.py:154)
  * SUM  of ||viewspace_points.grad[:, :2]|| where visible and of the visibility indicator
         -> xyz_gradient_accum, denom               (scene/gaussian_model.py:693-695)
  * MAX  of radii                                   -> max_radii2D (train.py:134)

How it is exchanged (`torch.distributed`; backend "nccl" == RCCL over xGMI on the GPU box, "gloo" in the CPU
tests).  xGMI is point-to-point (7 links per GPU), so a ring all-reduce is bound by one link.  Instead ONE flat
buffer per step -- the fp32 gradients and statistics followed by the int32 radii -- goes through
    all-to-all (every rank sends shard r of its buffer to rank r: 7 concurrent point-to-point transfers)
 -> local combine of the `world` received copies of the own shard, in rank order: SUM for the fp32 part, MAX for
    the radii tail (one small HIP kernel, r3dgs_reduce_shards; fixed order, so replicas stay bit-identical)
 -> all-gather of the combined shards.
Two collectives per step carry everything (the separate MAX all-reduce of round 1 is gone).  The exchange runs on
its own stream and the buffers are double-buffered: the rasterizer's backward of step k writes its gradients
straight into buffer k % 2 (`arena`), `exchange_async()` starts moving it, and step k+1's forward / backward
proceed meanwhile on buffer (k+1) % 2; `wait()` is called where the optimizer needs the sums.

Camera-sharded statistics of the pruning / culling passes (SURVEY.md 8e tier 2):
  * `merge_colour_variance`: per-rank partial results of calculate_colours_variance over disjoint camera subsets
    -> the statistics over all cameras (plain sums for the weights and distance accumulators, pairwise
    weighted-Welford merge for mean / variance; reduced_3dgs.cu:154-198);
  * `min_over_ranks`: element-wise MIN of find_minimum_projected_pixel_size over camera shards
    (redundancy_score.cu:95).
"""
import torch
import torch.distributed as dist


class ViewParallelExchange:
    def __init__(self, shapes, P, device, two_phase=True, buffers=2):
        """shapes: dict name -> per-Gaussian trailing shape of each gradient tensor, e.g.
        {"means3D": (3,), "sh": (16, 3), "opacity": (1,), "scales": (3,), "rotations": (4,)}."""
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        self.rank = dist.get_rank() if dist.is_initialized() else 0
        self.device = torch.device(device)
        self.shapes = {k: tuple(v) for k, v in shapes.items()}
        self.nbuf = max(1, int(buffers))
        self.two_phase = two_phase and self.world > 1
        self.on_gpu = self.device.type == "cuda"
        self.comm_stream = torch.cuda.Stream(device=device) if self.on_gpu else None
        self._pending = [None] * self.nbuf
        self._layout(P)

    def _layout(self, P):
        self.P = P
        self.slices = {}
        off = 0
        for name, shp in self.shapes.items():
            n = 1
            for s in shp:
                n *= s
            self.slices[name] = (off, off + P * n, (P,) + tuple(shp))
            off += P * n
        self.stat_off = off          # [P] grad-norm contributions, then [P] visibility counts
        self.sum_len = off + 2 * P   # fp32 (SUM) part; the int32 radii (MAX) follow
        total = self.sum_len + P
        pad = (-total) % max(self.world, 1)
        self.total = total
        self.shard = (total + pad) // max(self.world, 1)
        self.flats = [torch.zeros(total + pad, dtype=torch.float32, device=self.device) for _ in range(self.nbuf)]
        self.cur = 0
        if self.two_phase:   # receive / combine scratch, allocated once per layout
            self.recv = torch.empty(self.world * self.shard, dtype=torch.float32, device=self.device)
            self.mine = torch.empty(self.shard, dtype=torch.float32, device=self.device)

    def resize(self, P):
        """The Gaussian count changed (densification / pruning, train.py:132-147): every exchange in flight is waited
        for, then the buffers are laid out again for P Gaussians.  All ranks must call it with the same P (they do:
        densify and prune decisions are deterministic functions of the exchanged statistics)."""
        for k in range(self.nbuf):
            self.wait(k)
        if self.on_gpu:
            torch.cuda.current_stream(self.device).wait_stream(self.comm_stream)
        if P != self.P:
            self._layout(P)

    # ---- views of the current buffer -------------------------------------------------------------------------------
    @property
    def flat(self):
        return self.flats[self.cur]

    def _radii_view(self, k):
        return self.flats[k][self.sum_len:self.sum_len + self.P].view(torch.int32)

    @property
    def radii(self):
        return self._radii_view(self.cur)

    def arena(self, name, shape):
        """For diff_gaussian_rasterization._C.set_gradient_arena: the rasterizer's backward then writes the parameter
        gradients straight into the current exchange buffer and pack() has nothing to copy for them."""
        hit = self.slices.get(name)
        if hit is None or tuple(hit[2]) != tuple(shape):
            return None
        return self.flat[hit[0]:hit[1]].view(hit[2])

    def pack(self, grads, viewspace_grad, radii):
        """grads: dict name -> tensor (this rank's view). viewspace_grad: [P,3] grad of the means2D dummy."""
        flat = self.flat
        for name, (a, b, _shape) in self.slices.items():
            g = grads[name]
            if g.data_ptr() == flat.data_ptr() + 4 * a and g.is_contiguous():
                continue   # born in the buffer (arena)
            flat[a:b].copy_(g.reshape(-1))
        P, o = self.P, self.stat_off
        if self.on_gpu and viewspace_grad.is_contiguous() and radii.dtype == torch.int32:
            from diff_gaussian_rasterization import _C   # one fused launch instead of ~8 small torch kernels
            _C.pack_view_stats(viewspace_grad, radii, flat[o:o + P], flat[o + P:o + 2 * P], self.radii)
            return
        vis = radii > 0
        flat[o:o + P].copy_(torch.norm(viewspace_grad[:, :2], dim=-1) * vis)
        flat[o + P:o + 2 * P].copy_(vis.to(torch.float32))
        self.radii.copy_(radii)

    # ---- the exchange ----------------------------------------------------------------------------------------------
    def _combine(self):
        """local SUM / MAX of the `world` received copies of the own shard, rank order"""
        begin = self.rank * self.shard
        if self.on_gpu:
            from diff_gaussian_rasterization import _C
            _C.reduce_shards(self.recv, self.world, begin, self.sum_len, self.mine)
            return
        r = self.recv.view(self.world, self.shard)
        n_sum = min(max(self.sum_len - begin, 0), self.shard)
        acc = r[0, :n_sum].clone()
        for w in range(1, self.world):   # same order as the kernel
            acc += r[w, :n_sum]
        self.mine[:n_sum] = acc
        if n_sum < self.shard:
            self.mine[n_sum:].view(torch.int32).copy_(r[:, n_sum:].contiguous().view(torch.int32).view(self.world, -1).amax(0))

    def _run(self, k):
        flat = self.flats[k]
        if self.world == 1:
            return
        if self.two_phase:
            try:
                dist.all_to_all_single(self.recv, flat)
            except (RuntimeError, NotImplementedError) as e:   # a backend without all-to-all: plain all-reduces from now on
                import warnings
                warnings.warn(f"view-parallel exchange: all_to_all_single unavailable ({e}); using all_reduce")
                self.two_phase = False
                return self._run(k)
            self._combine()
            dist.all_gather_into_tensor(flat, self.mine)
        else:
            dist.all_reduce(flat[:self.sum_len], op=dist.ReduceOp.SUM)
            dist.all_reduce(self._radii_view(k), op=dist.ReduceOp.MAX)

    def exchange_async(self):
        """Starts the exchange of the current buffer on the communication stream and makes the NEXT buffer current
        (so that the following step's backward has somewhere to write).  Returns the index to pass to wait()."""
        k = self.cur
        if self.on_gpu and self.world > 1:
            main = torch.cuda.current_stream(self.device)
            self.comm_stream.wait_stream(main)          # gradients / statistics of this step are complete
            with torch.cuda.stream(self.comm_stream):
                self._run(k)
                ev = torch.cuda.Event()
                ev.record(self.comm_stream)
            self._pending[k] = ev
        else:
            self._run(k)
        self.cur = (self.cur + 1) % self.nbuf
        if self._pending[self.cur] is not None:         # the buffer about to be overwritten must have been consumed
            self.wait(self.cur)
        return k

    def wait(self, k):
        ev = self._pending[k]
        if ev is not None:
            torch.cuda.current_stream(self.device).wait_event(ev)
            self._pending[k] = None

    def exchange(self):
        """Synchronous form: exchange the current buffer and keep it current (round-1 behaviour)."""
        k = self.cur
        if self.on_gpu and self.world > 1:
            nxt = self.exchange_async()
            self.wait(nxt)
            self.cur = k
        else:
            self._run(k)

    def unpack(self, k=None):
        """-> (dict name -> summed gradient view, grad_norm_sum[P], visible_count[P], max_radii[P]) of buffer k
        (default: the current one)."""
        k = self.cur if k is None else k
        flat = self.flats[k]
        out = {name: flat[a:b].view(shape) for name, (a, b, shape) in self.slices.items()}
        P, o = self.P, self.stat_off
        return out, flat[o:o + P], flat[o + P:o + 2 * P], self._radii_view(k)


# ---- camera-sharded statistics (tier 2) -----------------------------------------------------------------
def merge_colour_variance_pair(a, b):
    """Pairwise merge of two PARTIAL results of calculate_colours_variance over disjoint camera sets.
    A partial is (accum[P,D], wSum[P,1], mean[P,1,3], S[P,1,3]) BEFORE the final divisions: accum = sum of w * colour
    distance, wSum = sum of w, mean = weighted mean of the full colour, S = the weighted sum of squared deviations the
    per-camera recurrence accumulates (reduced_3dgs.cu:154-198).  Weighted Welford / Chan et al.:
        w = wa + wb,  d = mean_b - mean_a,  mean = mean_a + d * wb / w,  S = Sa + Sb + d^2 * wa * wb / w.
    Exact for the textbook recurrence; the reference's `mean_old` aliasing (it squares the deviation from the UPDATED
    mean) makes its own S depend on the camera order, so a sharded run agrees with a sequential one up to that
    order dependence, not bit for bit."""
    acc_a, w_a, m_a, s_a = a
    acc_b, w_b, m_b, s_b = b
    w = w_a + w_b
    wa3, wb3, w3 = w_a.view(-1, 1, 1), w_b.view(-1, 1, 1), w.view(-1, 1, 1)
    frac = torch.where(w3 > 0, wb3 / w3, torch.zeros_like(w3))
    d = m_b - m_a
    mean = m_a + d * frac
    S = s_a + s_b + d * d * (wa3 * frac)
    return acc_a + acc_b, w, mean, S


def merge_colour_variance(partial, group=None):
    """All ranks contribute their partial (see merge_colour_variance_pair); every rank returns the FINAL
    (colour_distances[P,D], variance[P,1,3], mean[P,1,3]) over all cameras, as calculate_colours_variance would
    (reduced_3dgs.cu:202).  Partials are gathered and merged in rank order, so all replicas get identical bits."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if world > 1:
        flat = torch.cat([t.reshape(-1) for t in partial])
        gathered = [torch.empty_like(flat) for _ in range(world)]
        dist.all_gather(gathered, flat, group=group)
        parts = []
        for g in gathered:
            off, items = 0, []
            for t in partial:
                items.append(g[off:off + t.numel()].view(t.shape))
                off += t.numel()
            parts.append(tuple(items))
    else:
        parts = [tuple(partial)]
    acc, w, mean, S = parts[0]
    for p in parts[1:]:
        acc, w, mean, S = merge_colour_variance_pair((acc, w, mean, S), p)
    return acc / w, S / w.view(-1, 1, 1), mean


def min_over_ranks(values, group=None):
    """Element-wise MIN over ranks, in place (find_minimum_projected_pixel_size over camera shards: every rank runs the
    operator on its cameras -- unseen Gaussians keep the operator's 10000 -- and the minimum is global,
    redundancy_score.cu:95)."""
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(values, op=dist.ReduceOp.MIN, group=group)
    return values
