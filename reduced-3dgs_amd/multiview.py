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

Three opt-in transports cut the bytes a step puts on the links (round 6; DESIGN.md section 7 has the bytes / time model):
  * `set_degrees(degrees)`: SH BANDS BY DEGREE.  For every band b >= 1 only the rows of the Gaussians whose degree reaches it
    travel; the bands above a Gaussian's degree are exact zeros on every rank (the backward writes them as zeros) and the
    degrees are replicated, so every rank derives the same row lists locally.  Lossless, bit-identical to the dense form.
  * `sparse=True | "auto"`: VISIBLE-UNION rows only.  The ranks all-gather their visibility bitmaps (P / 8 bytes each), OR
    them, and exchange a compact buffer holding the rows of the Gaussians at least one rank saw -- every other row is an
    exact zero on every rank by the rasterizer's contract (gradients, statistics and radii of a culled Gaussian are
    written as zeros), so its sum is the zero the dense buffer already holds.  The combination order per element is
    the rank order in both forms: sparse == dense BIT FOR BIT.  Costs one small collective and one host read of the
    union's size per step ("auto": falls back to the dense form for the step when the union exceeds 85 % of P --
    eight unrelated cameras of one scene usually see nearly all of it; neighbouring cameras do not).
  * `sh_rest_bf16=True`: the 45 higher-band SH gradients of a Gaussian (3/4 of the buffer) travel as bfloat16 in both
    phases, accumulated in fp32 in rank order and rounded once (r3dgs_reduce_shards_mixed); every replica -- the owner of
    a shard included -- continues with the rounded sums, so replicas stay bit-identical.  248 -> 158 bytes per Gaussian.
    Not the reference's arithmetic (a relative 2^-8 per rounding on those gradients): opt-in.

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
    def __init__(self, shapes, P, device, two_phase=True, buffers=2, sparse=False, sh_rest_bf16=False,
                 sparse_threshold=0.85):
        """shapes: dict name -> per-Gaussian trailing shape of each gradient tensor, e.g.
        {"means3D": (3,), "sh": (16, 3), "opacity": (1,), "scales": (3,), "rotations": (4,)}.
        sparse: False (dense buffer, zero-copy with the gradient arena) | True (visible-union rows) | "auto" (visible-union
        rows when the union is below `sparse_threshold` of P, the dense form otherwise).  sh_rest_bf16: bands >= 1 of the
        "sh" gradient travel as bfloat16 (needs a tensor named "sh" of shape (M, 3), M > 1)."""
        assert sparse in (False, True, "auto")
        self.sparse = sparse
        self.sparse_threshold = float(sparse_threshold)
        self.sh_rest_bf16 = bool(sh_rest_bf16) and "sh" in shapes and len(shapes["sh"]) == 2 and shapes["sh"][0] > 1
        self.last = {"form": "dense", "rows": int(P), "bytes": None}   # what the last exchange put on the links
        self._deg, self._band_all = None, None                          # set_degrees()
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
        # compact transport (visible-union rows and / or bfloat16 SH bands): its own buffer, sized for all P rows once
        self._deg, self._band_all = None, None     # a new layout: the caller sets the degrees again (set_degrees)
        self.xbuf = None
        if self.compact:   # sized for the dense row count (every compact form is at most that + its padding)
            self.xbuf = torch.zeros(self.flats[0].numel() + self.world, dtype=torch.float32, device=self.device)
        w8 = torch.tensor([1, 2, 4, 8, 16, 32, 64, 128], dtype=torch.uint8, device=self.device)
        self._bit_weights = w8

    @property
    def compact(self):
        return bool(self.sparse) or self.sh_rest_bf16 or self._deg is not None

    def set_degrees(self, degrees):
        """Per-Gaussian SH degrees ([P] or [P, 1] int tensor, the SAME on every rank -- `GaussianModel._degrees` is replicated
        state) -> the exchange carries, for every SH band b >= 1, only the rows of the Gaussians whose degree reaches it:
        the bands above a Gaussian's degree are exact zeros on every rank by the rasterizer's contract (its backward writes
        them as zeros), so -- as with the visible-union rows -- the dense arena already holds their sum.  Lossless, bit-identical
        to the dense form; the mixed-degree scenes the method produces (reduced-3dgs culls SH bands per Gaussian) drop from
        248 to ~146 bytes per row at a uniform mix of degrees 0..3.  None switches it off.  Call it again whenever the degrees
        change (oneupSHdegree, cull_sh_bands) and after resize()."""
        if degrees is None or "sh" not in self.shapes or len(self.shapes["sh"]) != 2:
            self._deg, self._band_all = None, None
        else:
            d = degrees.reshape(-1).to(device=self.device, dtype=torch.int64)
            assert d.numel() == self.P, "set_degrees: one degree per Gaussian"
            self._deg = d
            nb = int(round(self.shapes["sh"][0] ** 0.5)) - 1          # highest band of the tensor (3 for M = 16)
            self._band_all = {b: torch.nonzero(d >= b).squeeze(1) for b in range(1, nb + 1)}
        if self.compact and self.xbuf is None:   # sized for the dense row count: whatever the degrees become, it fits
            self.xbuf = torch.zeros(self.flats[0].numel() + self.world, dtype=torch.float32, device=self.device)

    def _plan(self, idx, counts_only=None):
        """The compact exchange buffer of a step as a list of SEGMENTS -- (tensor, column range of its per-Gaussian row, row
        index list or None = all P rows, kind f32 | bf16 | i32) -- laid out in three regions of 4-byte words:
        [fp32 SUM | bfloat16 pairs SUM | int32 MAX].  idx: the visible-union rows (None: all).  counts_only: a dict
        {"rows": U, band: n_b} to lay out hypothetical counts (bytes_per_rank) without index lists."""
        segs = []
        U = (self.P if idx is None else int(idx.numel())) if counts_only is None else int(counts_only["rows"])

        def band_rows(b):
            if counts_only is not None:
                return None, int(counts_only.get(b, U))
            if self._deg is None:
                return idx, U
            if idx is None:
                r = self._band_all[b]
            else:
                r = idx[self._deg.index_select(0, idx) >= b]
            return r, int(r.numel())
        for name, shp in self.shapes.items():
            n = 1
            for s_ in shp:
                n *= s_
            if name == "sh" and len(shp) == 2 and shp[0] > 1 and (self.sh_rest_bf16 or self._deg is not None):
                ch = shp[1]
                kind = "bf16" if self.sh_rest_bf16 else "f32"
                segs.append(dict(name=name, c0=0, c1=ch, rows=idx, n=U, kind="f32"))          # the DC band: every row
                if self._deg is None:
                    segs.append(dict(name=name, c0=ch, c1=n, rows=idx, n=U, kind=kind))
                else:
                    nb = int(round(shp[0] ** 0.5)) - 1
                    for b_ in range(1, nb + 1):
                        r, cnt = band_rows(b_)
                        segs.append(dict(name=name, c0=ch * b_ * b_, c1=ch * (b_ + 1) * (b_ + 1), rows=r, n=cnt, kind=kind))
            else:
                segs.append(dict(name=name, c0=0, c1=n, rows=idx, n=U, kind="f32"))
        segs.append(dict(name="@grad_norm", c0=0, c1=1, rows=idx, n=U, kind="f32"))
        segs.append(dict(name="@visible", c0=0, c1=1, rows=idx, n=U, kind="f32"))
        segs.append(dict(name="@radii", c0=0, c1=1, rows=idx, n=U, kind="i32"))
        off = 0
        for sg in segs:
            if sg["kind"] == "f32":
                sg["off"] = off
                off += sg["n"] * (sg["c1"] - sg["c0"])
        sum_len, half = off, 0
        for sg in segs:
            if sg["kind"] == "bf16":
                sg["off"] = half                       # in halves, relative to the region
                half += sg["n"] * (sg["c1"] - sg["c0"])
        half_end = sum_len + (half + 1) // 2
        off = half_end
        for sg in segs:
            if sg["kind"] == "i32":
                sg["off"] = off
                off += sg["n"]
        total = off
        pad = (-total) % max(self.world, 1)
        return {"segs": segs, "rows": U, "sum_len": sum_len, "half": half, "half_end": half_end, "total": total,
                "padded": total + pad, "shard": (total + pad) // max(self.world, 1)}

    def bytes_per_rank(self, rows=None):
        """Bytes ONE rank sends per step: (world - 1) / world of the buffer in each of the two phases (all-to-all, then
        all-gather), plus -- sparse forms -- its visibility bitmap to every peer.  rows: the rows exchanged (default P; the
        per-band row counts of set_degrees are scaled with it)."""
        w = max(self.world, 1)
        if not self.compact:
            return int(2 * (w - 1) * self.shard * 4)
        U = self.P if rows is None else int(rows)
        counts = {"rows": U}
        if self._band_all is not None:
            for b_, r in self._band_all.items():
                counts[b_] = int(round(int(r.numel()) * U / max(self.P, 1)))
        x = self._plan(None, counts_only=counts)
        bitmap = (w - 1) * ((self.P + 7) // 8) if self.sparse else 0
        return int(2 * (w - 1) * x["shard"] * 4 + bitmap)

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

    # ---- compact transport ----------------------------------------------------------------------------------------
    def _staged(self):
        """GPU tensors over a backend that only moves host memory (the single-GPU gloo dry run of bench.py's N > 1 path,
        tools/dryrun_2rank.py): the two collectives of the compact form go through host copies.  Functional only."""
        return self.on_gpu and dist.get_backend() != "nccl"

    def _all_to_all(self, recv, send):
        if self._staged():
            r, s_ = recv.cpu(), send.cpu()
            dist.all_to_all_single(r, s_)
            recv.copy_(r)
        else:
            dist.all_to_all_single(recv, send)

    def _all_gather(self, out, inp):
        if self._staged():
            o, i_ = out.cpu(), inp.cpu()
            dist.all_gather_into_tensor(o, i_)
            out.copy_(o)
        else:
            dist.all_gather_into_tensor(out, inp)

    def _union_rows(self, k):
        """Indices of the Gaussians at least one rank saw this step (radii > 0 anywhere), ascending; None = all rows."""
        if not self.sparse:
            return None
        P = self.P
        vis = self._radii_view(k) > 0
        nb = (P + 7) // 8
        bits = torch.zeros(nb * 8, dtype=torch.uint8, device=self.device)
        bits[:P] = vis.to(torch.uint8)
        packed = (bits.view(nb, 8) * self._bit_weights).sum(dim=1, dtype=torch.int32).to(torch.uint8)
        if self.world > 1:
            allb = torch.empty(self.world * nb, dtype=torch.uint8, device=self.device)
            self._all_gather(allb, packed)
            allb = allb.view(self.world, nb)
            union = allb[0]
            for w in range(1, self.world):
                union = union | allb[w]
        else:
            union = packed
        mask = ((union.view(nb, 1) & self._bit_weights) != 0).view(-1)[:P]
        idx = torch.nonzero(mask).squeeze(1)          # (the one host read of the step: the union's size)
        if self.sparse == "auto" and idx.numel() > self.sparse_threshold * P:
            return None
        return idx

    def _dense2d(self, k, name):
        """[P, n] view of a tensor's (or a statistic's) part of dense buffer k."""
        flat, P, o = self.flats[k], self.P, self.stat_off
        if name == "@grad_norm":
            return flat[o:o + P].view(P, 1)
        if name == "@visible":
            return flat[o + P:o + 2 * P].view(P, 1)
        if name == "@radii":
            return self._radii_view(k).view(P, 1)
        a, b, _shape = self.slices[name]
        return flat[a:b].view(P, -1)

    def _pack_compact(self, k, idx):
        xb = self.xbuf
        x = self._plan(idx)
        hv = xb[x["sum_len"]:x["half_end"]].view(torch.bfloat16)
        for sg in x["segs"]:
            if sg["n"] == 0:
                continue
            src = self._dense2d(k, sg["name"])[:, sg["c0"]:sg["c1"]]
            r = (src if sg["rows"] is None else src.index_select(0, sg["rows"])).reshape(-1)
            if sg["kind"] == "f32":
                xb[sg["off"]:sg["off"] + r.numel()] = r
            elif sg["kind"] == "bf16":
                hv[sg["off"]:sg["off"] + r.numel()] = r.to(torch.bfloat16)
            else:
                xb[sg["off"]:sg["off"] + r.numel()].view(torch.int32).copy_(r)
        if x["half"] & 1:
            hv[x["half"]] = 0
        xb[x["total"]:x["padded"]] = 0
        return x

    def _unpack_compact(self, k, idx, x):
        xb = self.xbuf
        hv = xb[x["sum_len"]:x["half_end"]].view(torch.bfloat16)
        for sg in x["segs"]:
            if sg["n"] == 0:
                continue
            cols = sg["c1"] - sg["c0"]
            cnt = sg["n"] * cols
            if sg["kind"] == "f32":
                r = xb[sg["off"]:sg["off"] + cnt]
            elif sg["kind"] == "bf16":
                r = hv[sg["off"]:sg["off"] + cnt].to(torch.float32)
            else:
                r = xb[sg["off"]:sg["off"] + cnt].view(torch.int32)
            dst = self._dense2d(k, sg["name"])[:, sg["c0"]:sg["c1"]]
            if sg["rows"] is None:
                dst.copy_(r.view(sg["n"], cols))
            else:
                dst.index_copy_(0, sg["rows"], r.view(sg["n"], cols))

    def _combine_compact(self, recv, mine, x):
        begin, shard = self.rank * x["shard"], x["shard"]
        if self.on_gpu:
            from diff_gaussian_rasterization import _C
            _C.reduce_shards_mixed(recv, self.world, begin, x["sum_len"], x["half_end"], mine)
            return
        r = recv.view(self.world, shard)
        n_sum = min(max(x["sum_len"] - begin, 0), shard)
        n_half = min(max(x["half_end"] - begin, 0), shard)
        if n_sum:
            acc = r[0, :n_sum].clone()
            for w in range(1, self.world):   # same order as the kernel
                acc += r[w, :n_sum]
            mine[:n_sum] = acc
        if n_half > n_sum:   # bfloat16 pairs: fp32 accumulation in rank order, one rounding (to nearest even)
            h = r[:, n_sum:n_half].contiguous().view(torch.bfloat16).view(self.world, -1).to(torch.float32)
            acc = h[0].clone()
            for w in range(1, self.world):
                acc += h[w]
            mine[n_sum:n_half].view(torch.bfloat16).copy_(acc.to(torch.bfloat16))
        if n_half < shard:
            mine[n_half:].view(torch.int32).copy_(r[:, n_half:].contiguous().view(torch.int32).view(self.world, -1).amax(0))

    def _run_compact(self, k):
        idx = self._union_rows(k)
        if idx is None and not self.sh_rest_bf16 and self._deg is None:   # "auto" found (nearly) every row in the union: the dense form
            self.last = {"form": "dense (union above threshold)", "rows": self.P, "bytes": int(2 * (self.world - 1) * self.shard * 4)}
            return self._run_dense(k)
        x = self._pack_compact(k, idx)
        n, shard = x["padded"], x["shard"]
        send = self.xbuf[:n]
        recv = torch.empty(self.world * shard, dtype=torch.float32, device=self.device)
        mine = torch.empty(shard, dtype=torch.float32, device=self.device)
        self._all_to_all(recv, send)
        self._combine_compact(recv, mine, x)
        self._all_gather(send, mine)
        self._unpack_compact(k, idx, x)
        bitmap = (self.world - 1) * ((self.P + 7) // 8) if self.sparse else 0
        self.last = {"form": ("visible-union rows" if idx is not None else "all rows") +
                             (", SH bands by degree" if self._deg is not None else "") +
                             (", bf16 SH bands >= 1" if self.sh_rest_bf16 else ""),
                     "rows": x["rows"], "bytes": int(2 * (self.world - 1) * shard * 4 + bitmap)}

    def _run(self, k):
        if self.world > 1 and self.compact:
            return self._run_compact(k)
        return self._run_dense(k)

    def _run_dense(self, k):
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
                return self._run_dense(k)
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


# ---- bytes / time model of the exchange (bench.py prints it next to what it measures) -------------
XGMI_LINK_GBPS = 153.0        # 7 links x ~153 GB/s per GPU, taken per direction; a full mesh, one link per peer
COLLECTIVE_LATENCY_US = 25.0  # assumed launch + synchronisation cost of one RCCL collective on an 8-GPU node (not measured:
                              # no such node was available to any round of this project)


def exchange_model(P, world, step_ms, floats_per_row=59, union_frac=1.0, sh_rest_bf16=False, sparse=False,
                   sh_rest_floats=45.0, link_GBps=XGMI_LINK_GBPS, latency_us=COLLECTIVE_LATENCY_US):
    """Predicted cost of one step's exchange on a full xGMI mesh and the serialised speed-up it allows.
    Row = floats_per_row gradient floats + 2 statistics + 1 radius (4 bytes each; the 45 higher-band SH floats 2 bytes each
    with sh_rest_bf16; sh_rest_floats < 45: the MEAN number of higher-band SH floats a row carries when the bands travel by
    degree, set_degrees -- 3 ((deg + 1)^2 - 1) averaged over the Gaussians, 19.5 for a uniform mix of degrees 0..3).  Two phases (all-to-all, all-gather); in each a rank sends 1 / world of the buffer to every peer
    over that peer's own link, so a phase takes buffer / world / link bandwidth + one collective latency.  sparse adds the
    bitmap all-gather (P / 8 bytes to every peer) and moves union_frac x P rows.
    -> dict(bytes_per_row, buffer_bytes, bytes_per_link_per_phase, exchange_ms, speedup, efficiency)."""
    row = 4 * (floats_per_row + 3 - 45) + (2 if sh_rest_bf16 else 4) * sh_rest_floats
    rows = P * (union_frac if sparse else 1.0)
    buf = row * rows
    per_link = buf / world
    t_phase = per_link / (link_GBps * 1e9) * 1e3 + latency_us * 1e-3
    t = 2 * t_phase
    if sparse:
        t += (P / 8.0) / (link_GBps * 1e9) * 1e3 + latency_us * 1e-3
    speedup = world * step_ms / (step_ms + t) if world > 1 else 1.0
    return {"bytes_per_row": round(row, 1) if row != int(row) else int(row), "rows": int(rows), "buffer_bytes": int(buf), "bytes_per_link_per_phase": int(per_link),
            "exchange_ms": round(t, 4), "speedup": round(speedup, 2), "efficiency": round(speedup / world, 3),
            "assumes": f"{link_GBps:.0f} GB/s per link and direction, {latency_us:.0f} us per collective, full mesh"}


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
