"""Training path: one torch.autograd.Function for the whole CODONNet.

The reference has no explicit backward: a training step is plain autograd through
CODON_x4.py:66-132.  Here forward keeps every activation the backward needs (codon_amd.model
._forward_impl(save=...)) and backward is an explicit kernel schedule:
  * dL/dx of every MFMA conv = the forward conv kernel on PACK_DGRAD weights, with the ReLU mask of the
    tensor the gradient flows INTO fused in the epilogue (MASK_RELU) and fan-in fused as ACCUM_OUT;
  * dL/dw = codon_conv2d_wgrad, accumulating across the 5 / 3 loop iterations that share weights
    (CODON_x4.py:74,122); 16-bit: the 1x1 convs' dL/dw and masked dL/dx come from one pass (codon_conv1x1_bwd);
  * CAC gate backward = ops.cac_backward (4 kernels);
  * stem / head: stencil + 1-channel wgrad;
  * dL/d(input images) when x / y require grad (what the reference's autograd would return): the stems' 64 -> 1 dgrad
    through the head stencil, plus the identity path of the final residual add for x.
"""
from __future__ import annotations

import torch

import os as _os

from . import _lib as L
from . import ops
from .ops import Slice

# 16-bit: dL/dw and dL/dx of the three 128 -> 64 1x1 convs in one pass each (codon_conv1x1_bwd); 0 = two kernels (A/B)
FUSED_1X1_BWD = _os.environ.get("CODON_FUSED_1X1_BWD", "1") != "0"
# dL/d(fuse) of the fusion trunk collected in one pass over the four dL/d(f_i) (ops.ew_sum_mask); 0 = a copy and three
# read-modify-write passes (A/B)
SUM_GFUSE = _os.environ.get("CODON_SUM_GFUSE", "1") != "0"
# the ReLU mask of in2 applied by the last dgrad that accumulates into dL/d(in2) (CODON_CONV_MASK_SUM); 0 = a separate pass (A/B)
MASK_IN_EPILOGUE = _os.environ.get("CODON_MASK_IN_EPILOGUE", "1") != "0"
# 16-bit: the CAC gate backward without its apply pass -- dL/d(pre) is formed in the staging of the 1x1 backward that
# consumes it, dL/d(inputs) accumulates in the reduce pass (ops.cac_backward_fused); 0 = the four-kernel form (A/B)
FUSED_CAC_BWD = _os.environ.get("CODON_FUSED_CAC_BWD", "1") != "0"
# ... and dL/d(inputs) += dL/d(out_i) taken in the epilogue of the LAST dgrad conv that forms dL/d(out_i) (a conv5x5 64->64 in
# both streams: matrix-bound, the extra 2.5 GB ride along) instead of in the HBM-bound reduce pass (ops.conv2d_sum_into);
# 0 = in the reduce pass (A/B).  Bit-identical either way.
SUM_IN_DGRAD = _os.environ.get("CODON_SUM_IN_DGRAD", "1") != "0"
# 16-bit: dL/d(fuse) = (dL/df_3 + dL/df_2 + dL/df_1 + dL/df_0) * [fuse > 0] collected by the trunk's own input-gradient convs:
# dL/df_3's buffer is the running sum, the conv5x5 that completes dL/df_2 / dL/df_1 adds it there (ops.conv2d_sum_into), the
# two convs of iteration 0 accumulate straight into it and the last one applies the mask (CODON_CONV_MASK_SUM) -- no pass over
# the four tensors.  0 = ops.ew_sum_mask (A/B).  fuse = relu(conv7(..)), f_{i+1} = confuse_fuse(..) + fuse: CODON_x4.py:119-128
SUM_GFUSE_IN_DGRAD = _os.environ.get("CODON_SUM_GFUSE_IN_DGRAD", "1") != "0"

# every fixed-order reduction of the backward (56 weight-gradient reduces, 30 CAC parameter sums, 3 one-channel reduces) as
# ONE launch at the end (ops.DeferredReduce / codon_reduce_multi), bit-identical results; 0 = a small launch behind each
# producer (A/B)
DEFER_REDUCE = _os.environ.get("CODON_DEFER_REDUCE", "1") != "0"

# parameters in a fixed order: the flat gradient buffer of codon_amd.dist uses the same order
_CONVS = ["input", "conv_input", "conv1", "conv2", "conv3", "confuse", "input_c", "conv_input_c", "conv4", "conv5",
          "conv6", "confuse_c", "conv7", "conv8", "conv9", "conv10", "confuse_fuse", "conv11", "output"]


def used_parameters(model):
    """The 44 tensors that receive gradients, in state_dict order (attention_*5 are never used)."""
    ps = [(n + ".weight", getattr(model, n).weight) for n in _CONVS]
    for i in range(5):
        ac = getattr(model, f"attention_c{i}")
        ps += [(f"attention_c{i}.mlp.1.weight", ac.mlp[1].weight), (f"attention_c{i}.mlp.1.bias", ac.mlp[1].bias),
               (f"attention_c{i}.mlp.3.weight", ac.mlp[3].weight), (f"attention_c{i}.mlp.3.bias", ac.mlp[3].bias)]
    for i in range(5):
        ps.append((f"attention_s{i}.spatial.conv.weight", getattr(model, f"attention_s{i}").spatial.conv.weight))
    return ps


class _CodonFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, model, x, y, *params):
        save = {}
        with torch.no_grad():
            xf, yf = x.float().contiguous(), y.float().contiguous()
            out = model._forward_impl(xf, yf, save)
        ctx.model, ctx.saved, ctx.x, ctx.y = model, save, xf, yf
        ctx.in_dtype = x.dtype
        return out if x.dtype == torch.float32 else out.to(x.dtype)

    @staticmethod
    def backward(ctx, g_out):
        model, S, x, y = ctx.model, ctx.saved, ctx.x, ctx.y
        with torch.no_grad():
            need = (ctx.needs_input_grad[1], ctx.needs_input_grad[2])
            sink = _grad_sink(model) if all(ctx.needs_input_grad[3:]) else None
            grads = _backward_impl(model, S, x.contiguous(), y.contiguous(), g_out.contiguous(), input_grads=need, sink=sink)
        ctx.saved = None
        gx = grads["__input"].to(ctx.in_dtype) if need[0] else None
        gyy = grads["__input_c"].to(ctx.in_dtype) if need[1] else None
        if sink is not None:       # the gradients were ADDED into the parameters' .grad storage by the kernels themselves
            return (None, gx, gyy) + (None,) * len(sink)
        return (None, gx, gyy) + tuple(grads[n].to(p.dtype) for n, p in used_parameters(model))


def _grad_sink(model):
    """name -> the parameter's .grad tensor, when a codon_amd.dist.GradSync owns the gradients of this model AND the backward
    runs inside its direct_backward() context (gs.backward(loss)):
    every used parameter's .grad is then an fp32 view of ONE flat buffer, and the backward kernels ADD into those views
    directly (what autograd's AccumulateGrad would do with returned tensors, minus 44 add launches and their traffic).
    None (the ordinary route: gradients are returned to autograd) unless every view is in place."""
    gs = model.__dict__.get("_grad_sink")
    gs = gs() if gs is not None else None
    # opt-in per backward call: ctx.needs_input_grad is fixed at forward time, so without the context a
    # torch.autograd.grad(loss, [x]) or loss.backward(inputs=[x]) would add parameter gradients nobody asked for into .grad
    if gs is None or not gs.direct or gs._armed <= 0:
        return None
    sink = {}
    for (n, p), view_ptr in zip(gs.named, gs.view_ptrs):
        g = p.grad
        if g is None or g.dtype != torch.float32 or g.data_ptr() != view_ptr or not g.is_contiguous() or g.device != p.device:
            return None
        sink[n] = g
    return sink


def codon_apply(model, x, y):
    params = [p for _, p in used_parameters(model)]
    return _CodonFn.apply(model, x, y, *params)


def _backward_impl(model, S, x, y, gy, debug=None, input_grads=(False, False), sink=None):
    """sink: None = the 44 parameter gradients come back as fresh fp32 tensors (one flat allocation); a dict name -> fp32
    tensor = they are ADDED into those tensors (codon_amd.dist.GradSync's views of its flat all-reduce buffer)."""
    B, _, H, W = x.shape
    dev = x.device
    adt = model._act_dtype()                       # activation-gradient dtype follows the activations
    new = lambda c: ops.new_act(B, c, H, W, adt, dev)
    f32 = lambda t: t if t.dtype == torch.float32 else t.float()
    gy = f32(gy)
    named = used_parameters(model)
    if sink is not None:
        G = dict(sink)                             # parameter gradients: always fp32
    else:
        flat = torch.empty(sum(p.numel() for _, p in named), dtype=torch.float32, device=dev)
        G, off = {}, 0
        for n, p in named:
            G[n] = flat[off:off + p.numel()].view(p.shape)
            off += p.numel()
    add0 = sink is not None                        # the first contribution to a gradient adds (sink) or overwrites
    red = ops.DeferredReduce() if DEFER_REDUCE else None
    started = set()                                # immediate form: gradients that hold their first contribution

    def Pd(name):
        return model._packed(name, L.PACK_DGRAD)

    scratch = {}

    def restage(saved, xs: Slice, first, second):
        """`stage` = cat(relu(conv_a(x)), relu(conv_b(x))): the saved tensor, or -- model.set_recompute() -- the two
        sibling convs run again from the saved block input (same kernels, same packed weights: bit-identical)."""
        if saved is not None:
            return saved
        if "stage" not in scratch:            # one buffer for all 13 restage calls (setdefault would allocate each time)
            scratch["stage"] = new(128)
        buf = scratch["stage"]
        ops.conv2d(xs, model._packed(first[0]), Slice(buf, 0, 64), first[1], relu=True)
        ops.conv2d(xs, model._packed(second[0]), Slice(buf, 64, 64), second[1], relu=True)
        return buf

    fused1 = ops.is_c8(adt) and FUSED_1X1_BWD

    fused_cac = fused1 and FUSED_CAC_BWD

    def bwd1x1(name, xs: Slice, gs: Slice, gxs: Slice, gate=None, fcat_base=0):
        """both gradients of a 1x1 conv on a ReLU output xs: dW (accumulated over the shared uses) and gx = (W^T g) * [xs > 0];
        16-bit: one pass over xs and gs (codon_conv1x1_bwd), else the wgrad + the masked dgrad conv.  gate: gs is the block's
        dL/d(out) and the conv's output gradient is formed from it while staging (codon_conv1x1_bwd_gated)."""
        if not fused1:
            wgrad(name, xs, gs, 1)
            ops.conv2d(gs, Pd(name), gxs, 1, relu_mask=xs)
            return
        key = name + ".weight"
        kw = dict(defer=(red, key)) if red is not None else dict(accumulate=add0 or key in started)
        started.add(key)
        if gate is not None:
            ops.conv1x1_bwd_gated(xs, gs, Pd(name), gxs, G[key], gate, fcat_base, **kw)
        else:
            ops.conv1x1_bwd(xs, gs, Pd(name), gxs, G[key], **kw)

    def wgrad(name, xs: Slice, gs: Slice, k: int):
        key = name + ".weight"
        kw = dict(defer=(red, key)) if red is not None else dict(accumulate=add0 or key in started)
        started.add(key)
        ops.conv2d_wgrad(xs, gs, G[key], k, **kw)

    def wgrad1ch(name, a: Slice, img, flip: bool):
        key = name + ".weight"
        kw = dict(defer=(red, key)) if red is not None else dict(accumulate=add0)
        ops.conv1ch_wgrad(a, img, G[key], flip=flip, **kw)

    # ---- tail: y_hat = output(t11) + x ; t11 = relu(conv11(f3))                       :129-131
    t11, f_last = S["t11"], S["f_last"]
    g_t = new(64)
    ops.stencil_1to64(gy, f32(model.output.weight), Slice(g_t), flip=True, mask=Slice(t11))
    wgrad1ch("output", Slice(t11), gy, True)
    wgrad("conv11", Slice(f_last), Slice(g_t), 3)
    g_f = new(64)                                   # dL/df_3 (no ReLU on f)
    ops.conv2d(Slice(g_t), Pd("conv11"), Slice(g_f), 3)
    del g_t

    # ---- fusion trunk, iterations 2..0                                                  :122-128
    g_fs = [g_f]                                    # dL/df_3, dL/df_2, ...: f_{i+1} = confuse_fuse(..) + fuse feeds each into dL/dfuse
    g_r2, g_stage = new(128), new(128)
    sum_fuse = ops.is_c8(adt) and SUM_GFUSE_IN_DGRAD
    g_total = g_f                                   # sum_fuse: the running dL/d(fuse); its first term IS dL/df_3 (no copy)
    for i in (2, 1, 0):
        T = S[f"trunk{i}"]
        xin, r2 = T["x"], T["r2"]
        stage = restage(T["stage"], Slice(xin), ("conv8", 5), ("conv9", 3))
        bwd1x1("confuse_fuse", Slice(r2), Slice(g_f), Slice(g_r2))
        wgrad("conv10", Slice(stage), Slice(g_r2), 5)
        ops.conv2d(Slice(g_r2), Pd("conv10"), Slice(g_stage), 5, relu_mask=Slice(stage))
        wgrad("conv8", Slice(xin), Slice(g_stage, 0, 64), 5)
        wgrad("conv9", Slice(xin), Slice(g_stage, 64, 64), 3)
        if sum_fuse and i == 0:
            # dL/df_0 is needed nowhere but in the sum (f_0 is fuse itself): both convs accumulate into the running sum, the
            # last one masks it
            ops.conv2d(Slice(g_stage, 64, 64), Pd("conv9"), Slice(g_total), 3, accumulate=True)
            ops.conv2d(Slice(g_stage, 0, 64), Pd("conv8"), Slice(g_total), 5, accumulate=True, relu_mask=Slice(S["fuse"]),
                       mask_sum=True)
            break
        g_prev = new(64)
        if sum_fuse:      # the conv3x3 first: the conv5x5 (matrix-bound, HBM to spare) completes dL/df_i and adds it to the sum
            ops.conv2d(Slice(g_stage, 64, 64), Pd("conv9"), Slice(g_prev), 3)
            ops.conv2d_sum_into(Slice(g_stage, 0, 64), Pd("conv8"), Slice(g_prev), 5, Slice(g_total), accumulate=True)
        else:
            ops.conv2d(Slice(g_stage, 0, 64), Pd("conv8"), Slice(g_prev), 5)
            ops.conv2d(Slice(g_stage, 64, 64), Pd("conv9"), Slice(g_prev), 3, accumulate=True)
        g_f = g_prev
        g_fs.append(g_f)
    # f_0 is fuse itself: dL/dfuse = (dL/df_3 + dL/df_2 + dL/df_1 + dL/df_0) * [fuse > 0]; fuse = relu(conv7(oc4))   :119-128
    if sum_fuse:
        g_fuse = g_total
    elif SUM_GFUSE:
        g_fuse = new(64)
        ops.ew_sum_mask(Slice(g_fuse), [Slice(t) for t in g_fs], mask=Slice(S["fuse"]))
    else:
        g_fuse = g_fs[0].clone()
        ops.ew_add_mask(Slice(g_fuse), Slice(g_fs[1]))
        ops.ew_add_mask(Slice(g_fuse), Slice(g_fs[2]))
        ops.ew_add_mask(Slice(g_fuse), Slice(g_fs[3]), mask=Slice(S["fuse"]))
    del g_f, g_fs, g_total
    wgrad("conv7", Slice(S["oc"]), Slice(g_fuse), 3)
    g_oc = new(128)                                 # dL/d[out | out_c] of block 4
    ops.conv2d(Slice(g_fuse), Pd("conv7"), Slice(g_oc), 3)
    del g_fuse

    # ---- MC + CAC blocks 4..0                                                           :74-118
    in2 = S["in2"]
    g_pre2 = None if fused_cac else new(128)
    sum_in = fused_cac and SUM_IN_DGRAD             # dL/d(inputs) += dL/d(out_{i-1}) by block i's last dgrads
    # running dL/d[inputs | inputs_c].  sum_in: its first term IS dL/d(out_4) -- that buffer is kept (no copy) and block 4's
    # input gradients go to a fresh one
    g_in2 = g_oc if sum_in else new(128)
    for i in (4, 3, 2, 1, 0):
        Bk = S[f"blk{i}"]
        xin, r2, r2_c, pre2 = Bk["x"], Bk["r2"], Bk["r2_c"], Bk["pre2"]
        ac, asp = getattr(model, f"attention_c{i}"), getattr(model, f"attention_s{i}")
        if debug is not None:
            debug[f"g_oc{i}"] = g_oc.clone()
        gate = None
        ckeys = (f"attention_c{i}.mlp.1.weight", f"attention_c{i}.mlp.1.bias", f"attention_c{i}.mlp.3.weight",
                 f"attention_c{i}.mlp.3.bias", f"attention_s{i}.spatial.conv.weight")
        cdefer = dict(defer=(red, ckeys)) if red is not None else {}
        if fused_cac:
            # no apply pass, no g_pre2: the two 1x1 backward launches below read g_oc itself (depth half first; the depth
            # chain overwrites only channels 0..63 of g_oc / g_x before the colour chain reads 64..127)
            dw1, db1, dw2, db2, dws, gate = ops.cac_backward_fused(
                Slice(g_oc, 0, 64), Slice(g_oc, 64, 64), Slice(pre2, 0, 64), Slice(pre2, 64, 64), Bk["ch"], Bk["sp"],
                Bk["pooled"], Bk["pools"], f32(ac.mlp[1].weight), f32(ac.mlp[1].bias), f32(ac.mlp[3].weight),
                f32(asp.spatial.conv.weight), Slice(g_in2, 0, 64), Slice(g_in2, 64, 64),
                accumulate_in=(2 if sum_in else (i != 4)), **cdefer)
        else:
            dw1, db1, dw2, db2, dws = ops.cac_backward(
                Slice(g_oc, 0, 64), Slice(g_oc, 64, 64), Slice(pre2, 0, 64), Slice(pre2, 64, 64), Bk["ch"], Bk["sp"],
                Bk["pooled"], Bk["pools"], f32(ac.mlp[1].weight), f32(ac.mlp[1].bias), f32(ac.mlp[3].weight),
                f32(asp.spatial.conv.weight),
                Slice(g_pre2, 0, 64), Slice(g_pre2, 64, 64), Slice(g_in2, 0, 64), Slice(g_in2, 64, 64),
                accumulate_in=(i != 4), **cdefer)
        if red is None:             # immediate form (A/B): the five small results go to their places by five small copies / adds
            for key, t_ in zip(ckeys, (dw1, db1, dw2, db2, dws)):
                G[key].add_(t_.view_as(G[key])) if add0 else G[key].copy_(t_.view_as(G[key]))
        # block input gradient: blocks 1..4 read oc_{i-1}; block 0 reads in2 (accumulate there)
        if i > 0:
            # g_oc is dead after this block's cac_backward + 1x1 backward launches: reuse it (unless it just became g_in2)
            g_x, acc0 = (new(128) if (sum_in and i == 4) else g_oc), False
        else:
            g_x, acc0 = g_in2, True
        # depth stream: pre = confuse(r2); r2 = relu(conv3(stage)); stage = [relu(conv1(x)) | relu(conv2(x))]
        stage = restage(Bk["stage"], Slice(xin, 0, 64), ("conv1", 3), ("conv2", 5))
        if gate is not None:
            bwd1x1("confuse", Slice(r2), Slice(g_oc, 0, 64), Slice(g_r2), gate, 64)      # depth = Fcat channels 64..127
        else:
            bwd1x1("confuse", Slice(r2), Slice(g_pre2, 0, 64), Slice(g_r2))
        wgrad("conv3", Slice(stage), Slice(g_r2), 5)
        ops.conv2d(Slice(g_r2), Pd("conv3"), Slice(g_stage), 5, relu_mask=Slice(stage))
        wgrad("conv1", Slice(xin, 0, 64), Slice(g_stage, 0, 64), 3)
        wgrad("conv2", Slice(xin, 0, 64), Slice(g_stage, 64, 64), 5)
        # block 0 writes into dL/d[inputs | inputs_c] itself, which is masked by in2 = relu(..) next: the LAST gradient that
        # fans into each half applies that mask to the sum (CODON_CONV_MASK_SUM) -- no separate pass over the 128 channels
        last0 = dict(relu_mask=Slice(in2, 0, 64), mask_sum=True) if (i == 0 and MASK_IN_EPILOGUE) else {}
        last1 = dict(relu_mask=Slice(in2, 64, 64), mask_sum=True) if (i == 0 and MASK_IN_EPILOGUE) else {}
        ops.conv2d(Slice(g_stage, 0, 64), Pd("conv1"), Slice(g_x, 0, 64), 3, accumulate=acc0)
        if sum_in and i > 0:      # g_x = dL/d(out_{i-1}) is complete with this launch: it also goes into dL/d(inputs)
            ops.conv2d_sum_into(Slice(g_stage, 64, 64), Pd("conv2"), Slice(g_x, 0, 64), 5, Slice(g_in2, 0, 64), accumulate=True)
        else:
            ops.conv2d(Slice(g_stage, 64, 64), Pd("conv2"), Slice(g_x, 0, 64), 5, accumulate=True, **last0)
        # colour stream: stage_c = [relu(conv4(x_c)) 5x5 | relu(conv5(x_c)) 3x3]
        stage_c = restage(Bk["stage_c"], Slice(xin, 64, 64), ("conv4", 5), ("conv5", 3))
        if gate is not None:
            bwd1x1("confuse_c", Slice(r2_c), Slice(g_oc, 64, 64), Slice(g_r2), gate, 0)  # colour = Fcat channels 0..63
        else:
            bwd1x1("confuse_c", Slice(r2_c), Slice(g_pre2, 64, 64), Slice(g_r2))
        wgrad("conv6", Slice(stage_c), Slice(g_r2), 5)
        ops.conv2d(Slice(g_r2), Pd("conv6"), Slice(g_stage), 5, relu_mask=Slice(stage_c))
        wgrad("conv4", Slice(xin, 64, 64), Slice(g_stage, 0, 64), 5)
        wgrad("conv5", Slice(xin, 64, 64), Slice(g_stage, 64, 64), 3)
        # (the conv3x3 first, so that the stream's LAST input gradient is the matrix-bound conv5x5, as in the depth stream)
        ops.conv2d(Slice(g_stage, 64, 64), Pd("conv5"), Slice(g_x, 64, 64), 3, accumulate=acc0)
        if sum_in and i > 0:
            ops.conv2d_sum_into(Slice(g_stage, 0, 64), Pd("conv4"), Slice(g_x, 64, 64), 5, Slice(g_in2, 64, 64), accumulate=True)
        else:
            ops.conv2d(Slice(g_stage, 0, 64), Pd("conv4"), Slice(g_x, 64, 64), 5, accumulate=True, **last1)
        if i > 0:
            g_oc = g_x                              # dL/d[out | out_c] of block i-1
    del g_oc, g_pre2, g_r2, g_stage

    # ---- heads: in2 = [relu(conv_input(stem)) | relu(conv_input_c(stem_c))]            :68-72
    if not MASK_IN_EPILOGUE:
        ops.ew_add_mask(Slice(g_in2), None, mask=Slice(in2))
    g_s = new(64)
    for nm_in, nm_ci, st, img, off in (("input", "conv_input", S["stem"], x, 0),
                                       ("input_c", "conv_input_c", S["stem_c"], y, 64)):
        wgrad(nm_ci, Slice(st), Slice(g_in2, off, 64), 3)
        ops.conv2d(Slice(g_in2, off, 64), Pd(nm_ci), Slice(g_s), 3, relu_mask=Slice(st))
        wgrad1ch(nm_in, Slice(g_s), img, False)
        # dL/d(input image) when asked for: the 64 -> 1 dgrad of the stem is the head stencil with the flipped,
        # transposed kernel; x also feeds the final residual add (CODON_x4.py:67,131): + gy
        if input_grads[off // 64]:
            wt = f32(getattr(model, nm_in).weight).flip(2, 3).permute(1, 0, 2, 3).contiguous()   # (1,64,3,3)
            g_img = torch.empty_like(gy)
            ops.head(Slice(g_s), wt, gy if off == 0 else torch.zeros_like(gy), g_img)
            G["__input_c" if off else "__input"] = g_img
    if red is not None:
        red.run(G, accumulate=add0)
    return G
