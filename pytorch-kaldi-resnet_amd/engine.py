"""Execution engine: sequences the libspkhip kernels for the forward, backward and inference passes of
NeuralSpeakerModel.  Pure orchestration - every tensor op is a HIP kernel launch from ops.py.

Data flow of one BasicBlock in training (reference scripts/model.py:48-64), NHWC fp32:
    raw1 = conv1(x)            [stats in the conv epilogue]    -> bn1 finalize (mean/invstd/scale/shift)
    raw2 = conv2(relu(bn1(raw1)))   bn1+relu fused into conv2's input staging, never materialised
    out  = relu(bn2(raw2) + shortcut)                          one fused elementwise pass
so each block costs 7 tensor passes over HBM instead of 11.  Saved for backward: x, raw1, raw2, out (+ the
downsample's raw output).  Backward recomputes relu(bn1(raw1)) on the fly inside wgrad's input staging.
"""
import torch

from . import ops
from .hip import MASK_ACT, MASK_BITS, MASK_NONE, MASK_RAW


class _Conv:
    def __init__(self, holder):
        self.h = holder
        self.cin, self.cout, self.k, self.stride = holder.cin, holder.cout, holder.k, holder.stride
        self.wpk = None
        self.wpk_t = None

    def repack(self, need_t):
        w = self.h.weight.data
        if self.cin == 1:
            return
        self.wpk = ops.pack_conv_weight(w, False, self.wpk)
        if need_t:
            self.wpk_t = ops.pack_conv_weight(w, True, self.wpk_t)

    def pack_jobs(self, need_t):
        """(weight, packed buffer, transpose) entries for the batched pack; allocates the packed buffers on first use."""
        w = self.h.weight.data
        if self.cin == 1:
            return []
        n = ops.packed_numel(w)
        if self.wpk is None or self.wpk.numel() != n:
            self.wpk = torch.empty(n, device=w.device, dtype=torch.float32)
        jobs = [(w, self.wpk, False)]
        if need_t:
            nt = ops.packed_numel(w, bwd=True)
            if self.wpk_t is None or self.wpk_t.numel() != nt:
                self.wpk_t = torch.empty(nt, device=w.device, dtype=torch.float32)
            jobs.append((w, self.wpk_t, True))
        return jobs


class _BN:
    def __init__(self, holder):
        self.h = holder
        self.c = holder.c
        self.t4 = None      # train: [mean, invstd, scale, shift] of the forward whose backward is pending
        self.s4 = None      # the same rows for a training-mode forward WITHOUT a backward (no-grad / predict in train mode)
        self.e2 = None      # eval:  [scale, shift]

    def finalize(self, partial, count, amax_in=None, est_out=None, keep=True):
        """keep=False: the statistics go to a scratch row set, so that a no-grad training-mode forward between forward_train
        and backward does not overwrite what the pending backward reads (the running statistics still advance, as in the
        reference: any train-mode forward updates them)."""
        name = "t4" if keep else "s4"
        if getattr(self, name) is None:
            setattr(self, name, torch.empty(4, self.c, device=partial.device, dtype=torch.float32))
        t4 = getattr(self, name)
        h = self.h
        ops.bn_finalize(partial, count, h.weight.data, h.bias.data, h.running_mean, h.running_var, h.num_batches_tracked,
                        t4, amax_in=amax_in, est_out=est_out)
        return t4

    def eval_coeffs(self):
        h = self.h
        if self.e2 is None:
            self.e2 = torch.empty(2, self.c, device=h.weight.device, dtype=torch.float32)
        ops.bn_eval_coeffs(h.weight.data, h.bias.data, h.running_mean, h.running_var, self.e2)
        return self.e2


class _Block:
    def __init__(self, bp):
        self.kind = bp.kind
        self.convs = [_Conv(bp.conv1), _Conv(bp.conv2)] + ([_Conv(bp.conv3)] if bp.kind == "bottleneck" else [])
        self.bns = [_BN(bp.bn1), _BN(bp.bn2)] + ([_BN(bp.bn3)] if bp.kind == "bottleneck" else [])
        self.ds = None
        if bp.downsample is not None:
            self.ds = (_Conv(bp.downsample[0]), _BN(bp.downsample[1]))


class _LogitsFn(torch.autograd.Function):
    """Bridges torch autograd to the hand-written backward: one node for the whole network.  Parameter
    gradients are written straight into the flat gradient arena (the .grad views); `anchor` only makes the
    output require grad."""

    @staticmethod
    def forward(ctx, eng, x, y, anchor):
        logits, saved = eng.forward_train(x, y)
        ctx.eng = eng
        ctx.saved = saved
        return logits

    @staticmethod
    def backward(ctx, dlogits):
        ctx.eng.backward(ctx.saved, dlogits.contiguous())
        ctx.saved = None
        return None, None, None, None


class Engine:
    def __init__(self, model):
        self.m = model
        r = model.res
        self.stem_conv = _Conv(r.conv1)
        self.stem_bn = _BN(r.bn1)
        self.blocks = []
        for li in range(1, 5):
            for bp in getattr(r, "layer%d" % li):
                self.blocks.append(_Block(bp))
        self.head_bn = _BN(model.bn1) if hasattr(model, "bn1") else None
        self.pool_mode = 1 if model.pooling == "mean+std" else 0
        self.dirty = True
        self._packed_for_bwd = False
        self._pack_tables = {}
        self._amax_pool = None
        self._amax_fwd = None          # slots of the TRAINING forward: they live on in the saved records until its backward
        self._amax_eval = None         # slots of eval-mode / no-grad forwards (never referenced by a backward)
        self._fwd_gen = 0              # generation of the training table: a backward refuses records of an older forward
        # weight gradients (compute-bound, off the critical path) run on a side stream so that they overlap the
        # HBM-bound BatchNorm-backward passes of the main dgrad chain
        self.wgrad_stream = None
        self.use_side_stream = True
        # BatchNorm-backward reductions come out of the producing data-gradient epilogue (EPI_BNBWD) where possible
        self.fuse_bn_reduce = True
        # ... and the BatchNorm-backward apply runs inside the stride-1 data gradients' input staging (IN_BNBWD)
        self.fuse_bn_apply = True
        # the forward BN+residual+ReLU pass also emits 1-bit sign masks of the block outputs; the fused backward kernels read
        # those instead of the activated tensors
        import os
        self.use_sign_masks = os.environ.get("SPK_SIGN_MASKS", "1") == "1"
        self.fuse_apply_min_c = int(os.environ.get("SPK_FUSE_APPLY_MINC", "0"))   # experiment knob: skip the fusion below C channels
        # ... and above.  f16x3 default: the fusion only on the 32-channel layer, which is HBM-bound (the fusion saves one tensor
        # read there); from 64 channels on the BatchNorm backward runs as its own pass that writes the gradient ONCE as an f16
        # pair tensor, and the data gradient (in-wave pipelined kernel) and the weight gradient stage it by plain copy
        # (None = by operand mode: 32 in the f16x3 mode with pair tensors, no limit otherwise; SPK_FUSE_APPLY_MAXC overrides)
        self._fuse_apply_max_c = int(os.environ["SPK_FUSE_APPLY_MAXC"]) if "SPK_FUSE_APPLY_MAXC" in os.environ else None
        # opt-in: 1x1 convolutions (Bottleneck blocks) keep the fusion at every width.  They are HBM-bound and the fusion saves a
        # tensor pass, but the fused 1x1 kernel is VALU-bound: ResNet-101 with one chunk length per step in [200, 400], same box:
        # 105.7-106.2 ms without it, 110.6-110.9 ms with it, 112.3 ms with every convolution fused (profiles/r03_c4_policy.log)
        self.fuse_apply_1x1 = os.environ.get("SPK_FUSE_APPLY_1X1", "0") == "1"
        # gradients wrt raw conv outputs travel as f16 pair tensors (include/spkhip.h) in the f16x3 mode
        self.pair_draw = os.environ.get("SPK_PAIR_DRAW", "1") == "1"
        # diagnostics (tests / bench, never the timed path): when a [4] int64 device tensor, every tensor that an f16x3
        # matrix-core kernel stages is also run through spk_f16_window_count under the scale slot its consumer uses
        # The blocks of the LAST stage sit between the pooling layer and the first BatchNorm backward + convolution that bound the
        # gradient's range again: sqrt'(mean) of scripts/model.py:453 is unbounded (1e8 x the rest for a row mean of 1e-20) and
        # travels down the identity shortcuts as dout.  There the BatchNorm-backward scale bound pairs every channel's own absmax
        # of dz with its own gamma*invstd (spk_bn_bwd_reduce chan_amax) - a tensor-wide absmax times the layer's largest |k1|
        # overshoots the values by the ratio of the two and pushes the whole pair tensor out of its fp16 window.  Costs the
        # stand-alone reduction pass for two more 100 MB tensors per step (their statistics no longer come from an epilogue).
        self.chan_amax = os.environ.get("SPK_CHAN_AMAX", "1") == "1"
        self.window_counts = None
        self.bound_log = None          # diagnostics: when a list, (Cout, k, bound slot, true-absmax slot) of every BatchNorm-backward scale

    # ---- helpers ---------------------------------------------------------------------------------------
    @property
    def fuse_apply_max_c(self):
        if self._fuse_apply_max_c is not None:
            return self._fuse_apply_max_c
        return 32 if (self.pair_draw and ops.split_for(3, True) == 3) else 1000000

    @fuse_apply_max_c.setter
    def fuse_apply_max_c(self, v):
        self._fuse_apply_max_c = v

    def _all_convs(self):
        for b in self.blocks:
            for c in b.convs:
                yield c
            if b.ds is not None:
                yield b.ds[0]

    def _repack(self, need_t):
        if not self.dirty and (self._packed_for_bwd or not need_t):
            return
        # one launch packs every convolution (forward order, plus the transposed order the backward needs); the job
        # table lives on the device and is rebuilt only when a buffer, the operand mode or need_t changes
        jobs = [j for c in self._all_convs() for j in c.pack_jobs(need_t)]
        key = [(w.data_ptr(), p.data_ptr(), t, ops.split_for(w.shape[2], t)) for w, p, t in jobs]
        tab = self._pack_tables.get(need_t)
        if tab is None or tab.key != key:
            assert not torch.cuda.is_current_stream_capturing() or tab is None, "pack table changed during graph capture"
            tab = self._pack_tables[need_t] = ops.PackTable(jobs, jobs[0][0].device)
        tab.run()
        self.dirty = False
        self._packed_for_bwd = need_t

    def _check_input(self, x):
        if not x.is_cuda:
            raise RuntimeError("pytorch_kaldi_resnet_amd runs on MI355X only: input must be a device tensor "
                               "(there is no CPU fallback)")
        if x.dim() != 3 or x.size(1) != self.m.feat_dim:
            raise RuntimeError("expected input [B, %d, T], got %s" % (self.m.feat_dim, tuple(x.shape)))
        if x.dtype != torch.float32:
            raise RuntimeError("expected float32 input")
        return x.contiguous()

    # ---- inference trunk (eval-mode BN folded into conv epilogues) -----------------------------------------
    def _fwd_pool(self, device, train=False):
        """absmax slot table of a forward pass (f16x3 operand mode), reset at its start.  The training forward has a table of
        its own: its slots (convolution inputs, raw conv outputs) live on in the saved records for the backward pass, so an
        eval-mode or no-grad forward between forward_train and backward must not touch them (it takes the other table), and
        a SECOND training forward before the backward invalidates the records (generation check in Engine.backward)."""
        if ops.split_for(3) != 3 and ops.split_for(3, True) != 3:
            return None
        name = "_amax_fwd" if train else "_amax_eval"
        if getattr(self, name) is None:
            setattr(self, name, ops.AmaxPool(device))
        pool = getattr(self, name)
        pool.reset()
        return pool

    def _count(self, x, slot, affine=None, pairs=False):
        if self.window_counts is not None and slot is not None:
            ops.f16_window_count(x, slot, self.window_counts, affine=affine, pairs=pairs)

    def trunk_eval(self, x):
        self._repack(False)
        pool = self._fwd_pool(x.device)
        take = (lambda: pool.take()) if pool is not None else (lambda: None)
        f16 = ops.split_for(3) == 3
        e = self.stem_bn.eval_coeffs()
        a_amax = take()
        a, _ = ops.stem_fwd(x, self.stem_conv.h.weight.data, epi_affine=(e[0], e[1]), relu=True, amax_out=a_amax)
        for b in self.blocks:
            evs = [bn.eval_coeffs() for bn in b.bns]
            if b.ds is not None:
                ed = b.ds[1].eval_coeffs()
                res, _ = ops.conv_fwd(a, b.ds[0].wpk, b.ds[0].cout, 1, b.ds[0].stride, epi_affine=(ed[0], ed[1]),
                                      in_amax=a_amax if ops.split_for(1) == 3 else None)
            else:
                res = a
            h, h_amax = a, a_amax
            n = len(b.convs)
            for i, (c, ev) in enumerate(zip(b.convs, evs)):
                last = i == n - 1
                o_amax = take()
                h, _ = ops.conv_fwd(h, c.wpk, c.cout, c.k, c.stride, epi_affine=(ev[0], ev[1]),
                                    epi_add=res if last else None, relu=True,
                                    in_amax=h_amax if ops.split_for(c.k) == 3 else None, out_amax=o_amax)
                h_amax = o_amax
            a, a_amax = h, h_amax
        return a

    def embed_eval(self, x):
        feat = self.trunk_eval(x)
        pooled = ops.stats_pool_fwd(feat, self.pool_mode)
        return ops.linear_fwd(pooled, self.m.fc1.weight.data, self.m.fc1.bias.data)

    def predict(self, x):
        """scripts/model.py:402-409.  Uses BN batch statistics when the module is in train mode, exactly like
        the reference would; decode.py always calls it under model.eval()."""
        x = self._check_input(x)
        with torch.no_grad():
            if self.m.training:
                emb, _ = self._embed_train(x, save=False)
                return emb
            return self.embed_eval(x)

    # ---- heads ----------------------------------------------------------------------------------------------
    def _head_fwd(self, emb, y, train, save):
        m = self.m
        sv = {}
        h = emb
        if m.loss in ("softmax", "AAM-v1"):
            bn = self.head_bn
            if train:
                part = ops.bn_stats_partial(emb)
                t4 = bn.finalize(part, emb.shape[0], keep=save)
                h = ops.bn_apply(emb, t4[2], t4[3], relu=True)
            else:
                e2 = bn.eval_coeffs()
                h = ops.bn_apply(emb, e2[0], e2[1], relu=True)
            sv["h"] = h
        if m.loss == "softmax":
            logits = ops.linear_fwd(h, m.last.weight.data, m.last.bias.data)
        else:
            if y is None:
                raise RuntimeError("AAM heads need labels (scripts/model.py:483: label=None fails in the reference too)")
            y = y.contiguous()
            hn, hinv = ops.l2norm_fwd(h)
            wn, winv = ops.l2norm_fwd(m.last.weight.data)
            B, S, D = h.shape[0], wn.shape[0], wn.shape[1]
            cosv = ops.gemm(hn, wn, B, S, D, D, 1, 1, D)
            logits = ops.aam_margin_fwd(cosv, y, m.m, m.s)
            if save:
                sv.update(hn=hn, hinv=hinv, wn=wn, winv=winv, cosv=cosv, y=y)
        return logits, sv

    def _head_bwd(self, emb, sv, dlogits, acc):
        m = self.m
        if m.loss == "softmax":
            dh = ops.linear_bwd(sv["h"], m.last.weight.data, dlogits, m.last.weight.grad, m.last.bias.grad, accumulate=acc)
        else:
            dcos = ops.aam_margin_bwd(sv["cosv"], sv["y"], dlogits, m.m, m.s)
            B, S = dcos.shape
            D = sv["wn"].shape[1]
            dhn = ops.gemm(dcos, sv["wn"], B, D, S, S, 1, D, 1)
            dwn = ops.gemm(dcos, sv["hn"], S, D, B, 1, S, D, 1)
            dh = ops.l2norm_bwd(sv["hn"], sv["hinv"], dhn)
            ops.l2norm_bwd(sv["wn"], sv["winv"], dwn, out=m.last.weight.grad, accumulate=acc)
        if m.loss in ("softmax", "AAM-v1"):
            bn = self.head_bn
            dh = ops.bn_backward(dh, emb, sv["h"], bn.t4, bn.h.weight.data, bn.h.weight.grad, bn.h.bias.grad, MASK_ACT,
                                 accumulate=acc)
        return dh

    # ---- training forward -----------------------------------------------------------------------------------
    def _trunk_train(self, x, save):
        saved = {"x": x, "blocks": []}
        pool = self._fwd_pool(x.device, train=save)
        if save:
            self._fwd_gen += 1                 # this forward now owns the per-engine records (BN statistics rows, scale slots)
        saved["gen"] = self._fwd_gen
        take = (lambda: pool.take()) if pool is not None else (lambda: None)
        f16 = ops.split_for(3) == 3
        raw0, st = ops.stem_fwd(x, self.stem_conv.h.weight.data, stats=True)
        B, F, T, _ = raw0.shape
        t4 = self.stem_bn.finalize(st, B * F * T, keep=save)
        use_masks = save and self.use_sign_masks
        a_amax = take()
        a = ops.bn_apply(raw0, t4[2], t4[3], relu=True, mask=use_masks, amax_out=a_amax)
        a, amask = a if use_masks else (a, None)
        saved["raw0"] = raw0
        for b in self.blocks:
            # in_amax[i]: slot with the absmax (or its upper estimate) of the values conv_i STAGES - the block input itself,
            # or relu(bn(raw_{i-1})) recomputed in the staging - shared by the forward conv and its weight gradient
            rec = {"x": a, "xmask": amask, "in_amax": [], "raw_amax": []}
            raws = []
            h, aff, h_amax = a, None, a_amax           # h_amax: slot describing the values the next conv stages from h
            for c, bn in zip(b.convs, b.bns):
                rec["in_amax"].append(h_amax)
                raw_amax = take()
                rec["raw_amax"].append(raw_amax)       # absmax(raw): bounds |xhat| in the BatchNorm backward's operand scale
                if ops.split_for(c.k) == 3:
                    self._count(h, h_amax, affine=aff)
                raw, st = ops.conv_fwd(h, c.wpk, c.cout, c.k, c.stride, in_affine=aff, stats=True,
                                       in_amax=h_amax if ops.split_for(c.k) == 3 else None, out_amax=raw_amax)
                # the next conv stages relu(bn(raw)): its operand-scale bound comes out of the BatchNorm finalize
                est = take()
                t4 = bn.finalize(st, raw.shape[0] * raw.shape[1] * raw.shape[2], amax_in=raw_amax, est_out=est, keep=save)
                raws.append(raw)
                h, aff, h_amax = raw, (t4[2], t4[3]), est
            out_amax = take()
            if b.ds is not None:
                rawd, st = ops.conv_fwd(a, b.ds[0].wpk, b.ds[0].cout, 1, b.ds[0].stride, stats=True,
                                        in_amax=a_amax if ops.split_for(1) == 3 else None)
                td = b.ds[1].finalize(st, rawd.shape[0] * rawd.shape[1] * rawd.shape[2], keep=save)
                out = ops.bn_apply(h, aff[0], aff[1], res=rawd, res_affine=(td[2], td[3]), relu=True, mask=use_masks,
                                   amax_out=out_amax)
                rec["rawd"] = rawd
            else:
                out = ops.bn_apply(h, aff[0], aff[1], res=a, relu=True, mask=use_masks, amax_out=out_amax)
            out, amask = out if use_masks else (out, None)
            rec["raws"] = raws
            rec["out"] = out
            rec["mask"] = amask
            if save:
                saved["blocks"].append(rec)
            a, a_amax = out, out_amax
        return a, saved

    def _embed_train(self, x, save):
        self._repack(save)
        feat, saved = self._trunk_train(x, save)
        pooled = ops.stats_pool_fwd(feat, self.pool_mode)
        emb = ops.linear_fwd(pooled, self.m.fc1.weight.data, self.m.fc1.bias.data)
        saved["feat"], saved["pooled"], saved["emb"] = feat, pooled, emb
        return emb, saved

    def forward_train(self, x, y):
        emb, saved = self._embed_train(x, True)
        logits, sv = self._head_fwd(emb, y, True, True)
        saved["head"] = sv
        return logits, saved

    def forward_logits(self, x, y=None):
        x = self._check_input(x)
        m = self.m
        if m.training and torch.is_grad_enabled():
            anchor = m.fc1.bias
            return _LogitsFn.apply(self, x, y, anchor)
        with torch.no_grad():
            if m.training:
                emb, _ = self._embed_train(x, save=False)
                logits, _ = self._head_fwd(emb, y, True, False)
            else:
                emb = self.embed_eval(x)
                logits, _ = self._head_fwd(emb, y, False, False)
            return logits

    # ---- side-stream weight gradients ----------------------------------------------------------------------------
    def _wgrad(self, x, dy, dw, k, stride, in_affine=None, accumulate=False, dy_amax=None, x_amax=None, dy_presplit=False):
        if not self.use_side_stream:
            return ops.conv_wgrad(x, dy, dw, k, stride, in_affine=in_affine, accumulate=accumulate, dy_amax=dy_amax,
                                  x_amax=x_amax, dy_presplit=dy_presplit)
        if self.wgrad_stream is None:
            self.wgrad_stream = torch.cuda.Stream()
        main = torch.cuda.current_stream()
        side = self.wgrad_stream
        side.wait_stream(main)                      # dy / x are produced on the main stream
        # keep the allocator from recycling them under the side kernel - ALSO inside a graph capture: a block freed during the
        # capture goes back to the graph's private pool and is handed to the next allocation on the main branch, while the side
        # branch may replay later (measured: SPK_GRAPH_SIDE=1 without this gave a different loss on every run)
        for t in (x, dy):
            t.record_stream(side)
        with torch.cuda.stream(side):
            ops.conv_wgrad(x, dy, dw, k, stride, in_affine=in_affine, accumulate=accumulate, dy_amax=dy_amax, x_amax=x_amax,
                           dy_presplit=dy_presplit)

    def _join_wgrad(self):
        if self.use_side_stream and self.wgrad_stream is not None:
            torch.cuda.current_stream().wait_stream(self.wgrad_stream)

    # ---- backward ---------------------------------------------------------------------------------------------
    def backward(self, saved, dlogits, on_stage_done=None):
        """Hand-written backward of forward_train.  Parameter gradients go to the .grad arena views (overwritten
        when the views are fresh, accumulated otherwise).  on_stage_done(name) is called after the last weight
        gradient of each ResNet stage has been enqueued (hook for overlapping the gradient all-reduce)."""
        m = self.m
        acc = not m.attach_grads()
        if saved.get("gen") != self._fwd_gen:
            raise RuntimeError("backward of a training forward whose BatchNorm statistics rows and operand-scale slots were reused "
                               "by a later training forward (the engine keeps one set per model: run backward before the next "
                               "forward that requires grad)")
        with torch.no_grad():
            demb = self._head_bwd(saved["emb"], saved["head"], dlogits, acc)
            dpooled = ops.linear_bwd(saved["pooled"], m.fc1.weight.data, demb, m.fc1.weight.grad, m.fc1.bias.grad,
                                     accumulate=acc)
            # f16x3 operand mode of the backward kernels: every gradient tensor that feeds a matrix-core stage travels with
            # a slot holding the float bits of its absmax (atomicMax'ed by the kernel that writes it); the consumer derives
            # its power-of-two operand scale from it.  Slots are taken in a fixed order from a table zeroed once per step.
            amx = None
            if ops.split_for(3, True) == 3:
                if self._amax_pool is None:
                    self._amax_pool = ops.AmaxPool(dpooled.device)
                amx = self._amax_pool
                amx.reset()
            d_amax = amx.take() if amx else None
            d = ops.stats_pool_bwd(saved["feat"], dpooled, self.pool_mode, amax_out=d_amax)
            if on_stage_done:
                on_stage_done("head")
            nblk = len(self.blocks)
            stage_of = []
            for li in range(1, 5):
                stage_of += [li] * len(getattr(m.res, "layer%d" % li))
            part = None      # BatchNorm-backward partial sums of `d`, when the producing dgrad epilogue made them
            last_stage = stage_of[-1]
            for bi in range(nblk - 1, -1, -1):
                # the gradient this block hands down is the gradient wrt the previous block's (or the stem's)
                # BN+ReLU output: let the data-gradient epilogue reduce that BatchNorm's backward statistics
                if bi > 0:
                    prev = (saved["blocks"][bi - 1]["raws"][-1], self.blocks[bi - 1].bns[-1].t4)
                else:
                    prev = (saved["raw0"], self.stem_bn.t4)
                chan = self.chan_amax and amx is not None and stage_of[bi] == last_stage
                if chan and bi > 0 and stage_of[bi - 1] == last_stage:
                    prev = None            # that BatchNorm's statistics come from its own reduction pass, with per-channel absmax
                d, part, d_amax = self._block_bwd(self.blocks[bi], saved["blocks"][bi], d, acc, part, prev, d_amax, amx, chan)
                saved["blocks"][bi] = None
                if on_stage_done and (bi == 0 or stage_of[bi - 1] != stage_of[bi]):
                    self._join_wgrad()
                    on_stage_done("layer%d" % stage_of[bi])
            # stem: d is the gradient wrt relu(bn(raw0))
            bn = self.stem_bn
            draw0 = ops.bn_backward(d, saved["raw0"], None, bn.t4, bn.h.weight.data, bn.h.weight.grad, bn.h.bias.grad,
                                    MASK_RAW, draw_out=d, accumulate=acc, partial=part)
            ops.stem_wgrad(saved["x"], draw0, self.stem_conv.h.weight.grad, accumulate=acc)
            self._join_wgrad()
            if on_stage_done:
                on_stage_done("stem")

    def _block_bwd(self, b, rec, dout, acc, dout_partial=None, prev=None, dout_amax=None, amx=None, chan=False):
        """Backward of one residual block.  -> (gradient wrt the block input, BN-backward partial sums of it or None,
        absmax slot of it or None).

        Walks the convs last to first.  `g` is the gradient wrt the OUTPUT of bn_i (+ReLU): dout for the last BN (mask
        = block output > 0), the data gradient of conv_{i+1} for the inner ones (mask recomputed from raw_i).  For a
        stride-1 conv_i the BatchNorm backward of bn_i is applied inside the data-gradient kernel's input staging
        (IN_BNBWD): no separate apply pass; the kernel writes draw_i (for the weight gradient) and, for the last BN,
        dz (the shortcut gradient) as side products.  Stride-2 convs (parity-class launches) keep the separate pass.
        amx (f16x3 backward): the step's absmax slot table; g / draw / dx each carry a slot (see Engine.backward).
        chan: the last BatchNorm's scale bound uses the per-channel absmax of dz (Engine.chan_amax; dout_partial must be None)."""
        x, raws, out = rec["x"], rec["raws"], rec["out"]
        n = len(b.convs)
        g, g_part, g_amax, dz = dout, dout_partial, dout_amax, None
        take = (lambda: amx.take()) if amx is not None else (lambda: None)
        for i in range(n - 1, -1, -1):
            c, bn, raw = b.convs[i], b.bns[i], raws[i]
            last = i == n - 1
            inp = x if i == 0 else raws[i - 1]
            in_aff = None if i == 0 else (b.bns[i - 1].t4[2], b.bns[i - 1].t4[3])
            hw = (inp.shape[1], inp.shape[2])
            act = out if last else None
            f16 = amx is not None and ops.split_for(c.k, True) == 3     # data / weight gradients with fp16 two-term operands
            # what this conv's data gradient must also do in its epilogue
            add_dz, bnb = False, None
            if i > 0:
                if c.stride == 1 and self.fuse_bn_reduce:
                    bnb = (raws[i - 1], None, b.bns[i - 1].t4)        # statistics of bn_{i-1}'s backward
            elif b.ds is None:
                add_dz = True                                           # identity shortcut: dx = dgrad + dz
                if c.stride == 1 and self.fuse_bn_reduce and prev is not None:
                    bnb = (prev[0], x, prev[1])                         # statistics for the previous block's last BN
                    if rec.get("xmask") is not None:
                        bnb = bnb + (rec["xmask"],)                     # sign bits of x instead of x itself
            res_amax, draw_amax = take(), take()
            # f16x3: the gradient wrt the raw conv output (draw) travels as an f16 PAIR tensor - written once in the two-term
            # form under a rigorous bound known before it is computed, staged by plain copy in its data and weight gradients
            pairs = f16 and self.pair_draw
            raw_amax = rec["raw_amax"][i] if f16 else None
            if self.fuse_bn_apply and c.stride == 1 and (self.fuse_apply_min_c <= c.cout <= self.fuse_apply_max_c
                                                         or (c.k == 1 and self.fuse_apply_1x1)):
                ca = None
                if g_part is None:
                    ca = amx.take_n(c.cout) if (chan and last and f16) else None
                    g_part = ops.bn_bwd_partial(g, raw, act, bn.t4, MASK_ACT if last else MASK_RAW, chan_amax=ca)
                est = take() if f16 else None
                coef = ops.bn_bwd_coef(g_part, raw.numel() // raw.shape[-1], bn.h.weight.data, bn.t4, bn.h.weight.grad,
                                       bn.h.bias.grad, acc, amax_in=g_amax, raw_amax=raw_amax, est_out=est, chan_amax=ca)
                draw = torch.empty_like(raw)
                # the shortcut gradient dz = dout*[out > 0]: with an identity shortcut and sign masks it is never stored - the
                # first conv's data-gradient epilogue re-forms it from dout and the mask bits
                lazy_dz = b.ds is None and rec.get("mask") is not None and n > 1
                dzb = torch.empty_like(raw) if (last and not lazy_dz) else None
                inb = (raw, act, bn.t4, coef)
                if last and rec.get("mask") is not None:
                    inb = inb + (rec["mask"],)                          # sign bits of the block output instead of it
                if add_dz and lazy_dz:
                    res = ops.conv_dgrad(g, c.wpk_t, c.cin, c.k, 1, hw, add=dout, add_mask=rec["mask"], bn_bwd=bnb,
                                         in_bnbwd=inb, side=(draw, dzb), in_amax=est, out_amax=res_amax, side_amax=draw_amax,
                                         side_presplit=pairs)
                else:
                    res = ops.conv_dgrad(g, c.wpk_t, c.cin, c.k, 1, hw, add=dz if add_dz else None, bn_bwd=bnb,
                                         in_bnbwd=inb, side=(draw, dzb), in_amax=est, out_amax=res_amax, side_amax=draw_amax,
                                         side_presplit=pairs)
                if last:
                    dz = dzb
                draw_slot = est if pairs else draw_amax
                self._count(draw, draw_slot, pairs=pairs)
                if self.bound_log is not None and est is not None:
                    self.bound_log.append((c.cout, c.k, est, draw_amax))
            else:
                lazy_dz = False
                est = take() if pairs else None
                pair = (g_amax, raw_amax, est) if pairs else None
                if last:
                    # separate BatchNorm-backward pass (then a plain, pipelined data gradient).  With sign masks it reads the
                    # bits instead of the block output, and with an identity shortcut dz is not stored: the first conv's
                    # data-gradient epilogue re-forms it from dout and the bits (as in the fused form above)
                    bits = rec.get("mask")
                    lazy_dz = b.ds is None and bits is not None and n > 1
                    ca = amx.take_n(c.cout) if (chan and pairs and g_part is None) else None
                    draw = ops.bn_backward(g, raw, bits if bits is not None else out, bn.t4, bn.h.weight.data, bn.h.weight.grad,
                                           bn.h.bias.grad, MASK_BITS if bits is not None else MASK_ACT,
                                           dz_out=None if lazy_dz else g, accumulate=acc, partial=g_part, amax_out=draw_amax,
                                           pair=pair, chan_amax=ca)
                    dz = None if lazy_dz else g                         # (dout now holds dz)
                else:
                    draw = ops.bn_backward(g, raw, None, bn.t4, bn.h.weight.data, bn.h.weight.grad, bn.h.bias.grad, MASK_RAW,
                                           draw_out=g, accumulate=acc, partial=g_part, amax_out=draw_amax, pair=pair)
                draw_slot = est if pairs else draw_amax
                self._count(draw, draw_slot, pairs=pairs)
                if self.bound_log is not None and est is not None:
                    self.bound_log.append((c.cout, c.k, est, draw_amax))
                if add_dz and dz is None and rec.get("mask") is not None:
                    res = ops.conv_dgrad(draw, c.wpk_t, c.cin, c.k, c.stride, hw, add=dout, add_mask=rec["mask"], bn_bwd=bnb,
                                         in_amax=draw_slot if f16 else None, out_amax=res_amax, in_presplit=pairs)
                else:
                    res = ops.conv_dgrad(draw, c.wpk_t, c.cin, c.k, c.stride, hw, add=dz if add_dz else None, bn_bwd=bnb,
                                         in_amax=draw_slot if f16 else None, out_amax=res_amax, in_presplit=pairs)
            g, g_part = res if bnb is not None else (res, None)
            g_amax = res_amax
            self._wgrad(inp, draw, c.h.weight.grad, c.k, c.stride, in_affine=in_aff, accumulate=acc,
                        dy_amax=draw_slot if f16 else None, x_amax=rec["in_amax"][i] if f16 else None, dy_presplit=pairs)
        dx, part, dx_amax = g, g_part, g_amax
        if b.ds is not None:
            cd, bnd = b.ds
            f16d = amx is not None and ops.split_for(1, True) == 3
            drawd_amax = take() if f16d else None
            drawd = ops.bn_backward(dz, rec["rawd"], None, bnd.t4, bnd.h.weight.data, bnd.h.weight.grad, bnd.h.bias.grad,
                                    MASK_NONE, draw_out=dz, accumulate=acc, amax_out=drawd_amax)
            self._count(drawd, drawd_amax)
            self._wgrad(x, drawd, cd.h.weight.grad, 1, cd.stride, accumulate=acc, dy_amax=drawd_amax,
                        x_amax=rec["in_amax"][0] if f16d else None)
            # the 1x1 gradient lands on top of the 3x3 one: the same slot ends up >= the absmax of the sum's final values
            ops.conv_dgrad(drawd, cd.wpk_t, cd.cin, 1, cd.stride, (x.shape[1], x.shape[2]), out=dx, accumulate=True,
                           in_amax=drawd_amax, out_amax=dx_amax)
            part = None
        return dx, part, dx_amax

    # ---- fused training step (forward + CE + backward, no autograd graph) --------------------------------------
    def loss_and_grad(self, x, y, on_stage_done=None):
        """One iteration of scripts/train_resnet.py:316-327 without the autograd shell: returns
        (loss [1] device tensor, logits, rank [B] int32 for accuracy).  Gradients land in the .grad arena."""
        x = self._check_input(x)
        y = y.contiguous()
        with torch.no_grad():
            logits, saved = self.forward_train(x, y)
            loss_row, dl, rank = ops.softmax_ce(logits, y, grad_scale=1.0 / logits.shape[0])
            loss = ops.mean(loss_row)
        self.backward(saved, dl, on_stage_done)
        return loss, logits, rank


STAGE_ORDER = ["head", "layer4", "layer3", "layer2", "layer1", "stem"]     # the order backward finishes the stages in


class GraphedTrainStep:
    """weight re-pack + forward + CE + backward of one fixed-shape batch captured as hipGraphs (torch.cuda.CUDAGraph): one
    host launch per step instead of ~450, so a busy host cannot starve the GPU.  Inputs are copied into static buffers;
    gradients land in the model's flat gradient arena exactly as in the eager path (always overwritten).
    The optimizer step stays outside the graph (its learning rate changes per epoch).

    segmented=False (one GPU): ONE graph.
    segmented=True (data parallel): the step is cut where backward finishes a ResNet stage - six graphs sharing one
    memory pool, replayed back to back.  Between two replays the host enqueues that stage's gradient all-reduce on the
    communication stream (`on_stage_done(name)`, parallel.GradAllReducer), so the RCCL kernels of stage k overlap the
    data / weight gradients of stages k-1..stem exactly as in the eager path (reference: DDP's bucketed all-reduce
    overlapped with backward, scripts/train_resnet.py:183-185,327), with 6 host launches per step instead of ~450 and
    no collective inside a captured region."""

    def __init__(self, engine, batch, frames, warmup=2, side_stream=None, segmented=False, pool=None):
        """pool: memory pool of another GraphedTrainStep of the same engine (`other.pool()`): steps for different chunk
        lengths are never replayed concurrently, so they can share one pool - the cache of per-length graphs then costs
        the activation memory of the longest length, not the sum."""
        import os
        if side_stream is None:
            side_stream = os.environ.get("SPK_GRAPH_SIDE", "0") == "1"
        m = engine.m
        dev = m.flat_parameters().device
        self.eng = engine
        # measured on MI355X / ROCm 7.2: a captured two-branch graph (weight gradients on the side stream) replays no faster than the
        # same kernels captured on one stream (round 1: ~2 % slower; round 3: 52.5-52.8 ms both ways), so the graph is recorded
        # single-stream
        side, engine.use_side_stream = engine.use_side_stream, bool(side_stream)
        self.x = torch.zeros(batch, m.feat_dim, frames, device=dev)
        self.y = torch.zeros(batch, dtype=torch.long, device=dev)
        self.shape = (batch, m.feat_dim, frames)
        # warm-up outside capture: allocates the workspaces / BN buffers the graph will reuse, picks tiles.
        # It runs real steps on a dummy batch, so the BatchNorm running statistics are snapshotted and restored.
        saved_buffers = [b.clone() for b in m.buffers()]
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(warmup):
                for p in m.parameters():
                    p.grad = None
                engine.loss_and_grad(self.x, self.y)
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        for b, sb in zip(m.buffers(), saved_buffers):
            b.copy_(sb)
        for p in m.parameters():
            p.grad = None                      # captured backward overwrites (no accumulation)
        # The optimizer rewrites the weights between replays (scripts/train_resnet.py:316-328: forward runs on what
        # optimizer.step() just wrote), so the weight re-pack MUST be a node of the graph: the warm-up steps left the
        # engine clean, mark it dirty so that the capture records the batched pack launch (its device-resident job table
        # was built during warm-up and is kept alive by self.pack_table).
        engine.dirty = True
        self.segments = []                     # [(graph, stage name reported after it or None)]
        cap = {"g": None, "ctx": None}

        def begin():
            g = torch.cuda.CUDAGraph()
            kw = {"pool": self.segments[0][0].pool()} if self.segments else ({"pool": pool} if pool is not None else {})
            # thread_local: helper threads of other libraries (the RCCL watchdog) may touch the device during capture
            ctx = torch.cuda.graph(g, capture_error_mode="thread_local", **kw)
            ctx.__enter__()
            cap["g"], cap["ctx"] = g, ctx

        def end(name):
            cap["ctx"].__exit__(None, None, None)
            self.segments.append((cap["g"], name))
            cap["g"] = cap["ctx"] = None

        def cut(name):                         # Engine.backward reports: every gradient of stage `name` is enqueued
            end(name)
            if name != STAGE_ORDER[-1]:
                begin()

        begin()
        try:
            self.loss, self.logits, self.rank = engine.loss_and_grad(self.x, self.y, cut if segmented else None)
            if cap["ctx"] is not None:
                end(None)
        except BaseException:
            if cap["ctx"] is not None:
                cap["ctx"].__exit__(None, None, None)
            raise
        self.graph = self.segments[0][0]
        assert [n for _, n in self.segments] == (STAGE_ORDER if segmented else [None])
        self.pack_table = engine._pack_tables.get(True)
        assert self.pack_table is not None and self.pack_table.launches_captured >= 1, \
            "the captured training step does not contain the conv-weight re-pack"
        engine.dirty = True                    # the capture itself executed nothing
        engine.use_side_stream = side

    def matches(self, x):
        return tuple(x.shape) == self.shape

    def pool(self):
        return self.segments[0][0].pool()

    def __call__(self, x, y, on_stage_done=None):
        self.x.copy_(x, non_blocking=True)
        self.y.copy_(y, non_blocking=True)
        for g, name in self.segments:          # segment 0 re-packs the conv weights from the parameter arena first
            g.replay()
            if name is not None and on_stage_done is not None:
                on_stage_done(name)
        self.eng.dirty, self.eng._packed_for_bwd = False, True
        self.eng.m.attach_grads()
        return self.loss, self.logits, self.rank


class GraphedStepCache:
    """Variable-length training (BASELINE configs[3]; reference scripts/datasets.py:40-43,53-57: chunk lengths drawn in
    [min, max], here ONE length per batch, identical on every rank): a GraphedTrainStep per (batch, frames), captured on
    first use, all sharing ONE memory pool - steps of different lengths are never replayed concurrently, so the cache costs
    the activation memory of the longest length it has seen, not the sum - and at most `max_graphs` of them (least recently
    used dropped first).  Host cost per step after the first visit of a length: the replay launches, < 1 ms."""

    def __init__(self, engine, segmented=False, max_graphs=64, warmup=2):
        self.eng, self.segmented, self.max_graphs, self.warmup = engine, segmented, max_graphs, warmup
        self.steps = {}            # (batch, frames) -> GraphedTrainStep, in least-recently-used order
        self._pool_owner = None    # keeps the shared pool alive even after its first graph was dropped
        self.captures = 0

    def get(self, batch, frames):
        key = (int(batch), int(frames))
        st = self.steps.pop(key, None)
        if st is None:
            pool = self._pool_owner.pool() if self._pool_owner is not None else None
            st = GraphedTrainStep(self.eng, key[0], key[1], warmup=self.warmup, segmented=self.segmented, pool=pool)
            if self._pool_owner is None:
                self._pool_owner = st
            self.captures += 1
            while len(self.steps) >= self.max_graphs:
                old = next(iter(self.steps))
                if self.steps[old] is self._pool_owner:      # never drop the owner of the pool: re-queue it
                    self.steps[old] = self.steps.pop(old)
                    if len(self.steps) == 1:
                        break
                    continue
                del self.steps[old]
        self.steps[key] = st
        return st

    def __call__(self, x, y, on_stage_done=None):
        return self.get(x.shape[0], x.shape[2])(x, y, on_stage_done)

    def __len__(self):
        return len(self.steps)
