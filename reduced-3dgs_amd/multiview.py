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

Collectives are `torch.distributed` (backend "nccl" == RCCL over xGMI on the GPU box, "gloo" in the CPU
tests).  Two calls per step, on flat pre-allocated buffers: one fp32 SUM (reduce-scatter + all-gather so
that every one of the 7 point-to-point xGMI links of a rank carries 1/8 of the buffer concurrently, instead
of a ring bounded by a single link) and one int32 MAX.
"""
import torch
import torch.distributed as dist


class ViewParallelExchange:
    def __init__(self, shapes, P, device, two_phase=True):
        """shapes: dict name -> per-Gaussian trailing shape of each gradient tensor, e.g.
        {"means3D": (3,), "sh": (16, 3), "opacity": (1,), "scales": (3,), "rotations": (4,)}."""
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        self.rank = dist.get_rank() if dist.is_initialized() else 0
        self.P = P
        self.slices = {}
        off = 0
        for name, shp in shapes.items():
            n = 1
            for s in shp:
                n *= s
            self.slices[name] = (off, off + P * n, (P,) + tuple(shp))
            off += P * n
        self.stat_off = off          # [P] grad-norm contributions, then [P] visibility counts
        total = off + 2 * P
        pad = (-total) % max(self.world, 1)
        self.flat = torch.zeros(total + pad, dtype=torch.float32, device=device)
        self.total = total
        self.radii = torch.zeros(P, dtype=torch.int32, device=device)
        self.two_phase = two_phase and self.world > 1

    def arena(self, name, shape):
        """For diff_gaussian_rasterization._C.set_gradient_arena: the rasterizer's backward then writes the parameter
        gradients straight into this exchange buffer and pack() has nothing to copy for them."""
        hit = self.slices.get(name)
        if hit is None or tuple(hit[2]) != tuple(shape):
            return None
        return self.flat[hit[0]:hit[1]].view(hit[2])

    def pack(self, grads, viewspace_grad, radii):
        """grads: dict name -> tensor (this rank's view). viewspace_grad: [P,3] grad of the means2D dummy."""
        for name, (a, b, _shape) in self.slices.items():
            g = grads[name]
            if g.data_ptr() == self.flat.data_ptr() + 4 * a and g.is_contiguous():
                continue   # born in the buffer (arena)
            self.flat[a:b].copy_(g.reshape(-1))
        P, o = self.P, self.stat_off
        if self.flat.is_cuda and viewspace_grad.is_contiguous() and radii.dtype == torch.int32:
            from diff_gaussian_rasterization import _C   # one fused launch instead of ~8 small torch kernels
            _C.pack_view_stats(viewspace_grad, radii, self.flat[o:o + P], self.flat[o + P:o + 2 * P], self.radii)
            return
        vis = radii > 0
        self.flat[o:o + P].copy_(torch.norm(viewspace_grad[:, :2], dim=-1) * vis)
        self.flat[o + P:o + 2 * P].copy_(vis.to(torch.float32))
        self.radii.copy_(radii)

    def exchange(self):
        if self.world == 1:
            return
        if self.two_phase:
            shard = self.flat.numel() // self.world
            mine = torch.empty(shard, dtype=torch.float32, device=self.flat.device)
            dist.reduce_scatter_tensor(mine, self.flat, op=dist.ReduceOp.SUM)
            dist.all_gather_into_tensor(self.flat, mine)
        else:
            dist.all_reduce(self.flat, op=dist.ReduceOp.SUM)
        dist.all_reduce(self.radii, op=dist.ReduceOp.MAX)

    def unpack(self):
        """-> (dict name -> summed gradient view, grad_norm_sum[P], visible_count[P], max_radii[P])"""
        out = {name: self.flat[a:b].view(shape) for name, (a, b, shape) in self.slices.items()}
        P, o = self.P, self.stat_off
        return out, self.flat[o:o + P], self.flat[o + P:o + 2 * P], self.radii
