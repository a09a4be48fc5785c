"""Shared helpers of the GPU parity tests (not a test module).

Rounding discontinuity: y_hat = round(y - mu) + mu turns float noise into +-1 where y - mu sits within ~1e-6 of a
half-integer.  The helpers below make every oracle comparison UNCONDITIONAL: the HIP path's rounding decisions are
read back (a no-grad forward with ``keep``), the oracle adopts them (``round_override``), and the number of decisions
that differ from the oracle's own ("flips") is counted and bounded separately."""
import math

import torch


def hip_round_decisions(forward, P, x, nz, ny, **kw):
    """forward = icm_amd.models.wacnn_forward / stf_forward.  Returns ({"y": round(y-mu), "z": round(z-med)}, keep)
    with CPU tensors; kernels are deterministic, so the decisions equal those of any other run on the same inputs."""
    from icm_amd import engine as E
    keep = {}
    with torch.no_grad():
        forward(E.Tape(need_grad=False), P, x, nz, ny, keep=keep, **kw)
    med = P["entropy_bottleneck.quantiles"][:, 0, 1].reshape(1, -1, 1, 1)
    ro = {"y": torch.round(keep["y"] - keep["mu"]).cpu(), "z": torch.round(keep["z"] - med).cpu()}
    return ro, {k: v.detach().cpu() for k, v in keep.items()}


def count_flips(ro, dbg, sd):
    """rounding decisions of the HIP path that differ from the oracle's own (dbg = oracle ``_dbg`` of a run WITHOUT
    override is not needed: y - mu of the overridden run equals the free run up to the first flip; we compare against
    the overridden run's own y - mu, which is what the oracle would round)"""
    med = sd["entropy_bottleneck.quantiles"][:, :, 1:2].reshape(1, -1, 1, 1).detach()
    fy = (torch.round((dbg["y"] - dbg["mu"]).detach()) != ro["y"]).sum().item()
    fz = (torch.round((dbg["z"] - med).detach()) != ro["z"]).sum().item()
    return int(fy), int(fz)


def near_half(dbg, eps=1e-4):
    t = (dbg["y"] - dbg["mu"]).detach()
    return int(((t - torch.floor(t) - 0.5).abs() < eps).sum().item())


def grad_errors(hip, ref, names):
    """hip / ref: name -> gradient tensor (None = zero).  Returns (total_ref_norm, worst_l2_over_total, worst_elem,
    rows): rows = [(name, ||d||_2, ||ref||_2, max|d| / max(max|ref|, floor))] with floor = 1e-6 x the largest
    gradient entry of the model (only exactly-zero tensors fall under it); worst_elem = the largest 4th column --
    an element-wise measure, so a fault confined to one MFMA tile of one tensor is not averaged away."""
    rows = []
    tot2, gmax = 0.0, 0.0
    pairs = []
    for n in names:
        r = ref.get(n)
        h = hip.get(n)
        if r is None and h is None:
            continue
        r = torch.zeros_like(h, device="cpu") if r is None else r.detach().cpu()
        h = torch.zeros_like(r) if h is None else h.detach().cpu()
        pairs.append((n, h, r))
        gmax = max(gmax, r.abs().max().item())
    for n, h, r in pairs:
        d = (h.double() - r.double())
        rn = r.double().norm().item()
        rows.append((n, d.norm().item(), rn, d.abs().max().item() / max(r.abs().max().item(), 1e-6 * gmax)))
        tot2 += rn ** 2
    tot = math.sqrt(tot2)
    return tot, max(r[1] for r in rows) / tot, max(r[3] for r in rows), rows


def oracle_train_step(fwd, sd, x, noise, it, st, pnames, main, lmbda=0.0067, lr=1e-4, **kw):
    """one iteration of the reference loop (train.py:188-214) on the oracle; returns the loss dict"""
    from oracle import wacnn_oracle as O
    for n in pnames:
        sd[n].grad = None
    out = fwd(sd, x, noise, **kw)
    Lr = O.rd_loss(x, out, lmbda)
    Lr["loss"].backward()
    grads = [sd[n].grad if sd[n].grad is not None else torch.zeros_like(sd[n]) for n in main]
    raw = {n: g.clone() for n, g in zip(main, grads)}
    O.clip_grad_norm_(grads, 1.0)
    with torch.no_grad():
        for n, g in zip(main, grads):
            O.adam_step(sd[n], g, st[n][0], st[n][1], it, lr)
    aux = O.eb_aux_loss(sd)
    (gq,) = torch.autograd.grad(aux, [sd["entropy_bottleneck.quantiles"]])
    with torch.no_grad():
        q = "entropy_bottleneck.quantiles"
        O.adam_step(sd[q], gq, st[q][0], st[q][1], it, lr)
    Lr["raw_grads"] = raw
    Lr["out"] = out
    return Lr


def trainable(sd):
    s = {k: (v.clone().requires_grad_(True) if v.dtype.is_floating_point and v.numel() else v) for k, v in sd.items()}
    pnames = [k for k, v in s.items() if isinstance(v, torch.Tensor) and v.requires_grad and
              k.rsplit(".", 1)[-1] not in ("pedestal", "bound", "target", "scale_bound", "scale_table")]
    main = [n for n in pnames if not n.endswith(".quantiles")]
    st = {n: (torch.zeros_like(s[n]), torch.zeros_like(s[n])) for n in pnames}
    return s, pnames, main, st


def update_l2(P, s, sd0, pnames):
    num = sum(((P[n].detach().cpu() - s[n].detach()).double() ** 2).sum().item() for n in pnames)
    den = sum(((s[n].detach() - sd0[n]).double() ** 2).sum().item() for n in pnames)
    return math.sqrt(num / den)
