"""Independent float64 statement of the RNb-NeuS2 network and its parameter gradients, differentiated by PyTorch autograd.

TEST INFRASTRUCTURE. This file does not restate the reference's hand-written backward pass (nerf_network.h:257-452,
grid.h:366-883, fully_fused_mlp.cu:885-1142) -- it states only the FORWARD mathematics

    f_l(x)   multi-resolution trilinear hash-grid features            (grid.h:237-321)
    y        = W1 relu(W0 [x - 0.5 | f(x) | 0]),   sdf = y_0 + bias   (nerf_network.h:149-160, 225-230)
    n        = d sdf / d x                                            (nerf_network.h:163-189)
    r        = C2 relu(C1 relu(C0 [y | 0_16 | x | n | 0_10]))         (nerf_network.h:206-219)
    L        = sum_s  dout[0:3] . r[0:3] + dout[3] sdf + (dout[4:7] / B + dout[8:11]) . n + dout[7] var

and lets autograd produce dL/d(parameter), the second-order terms (n depends on W0, W1 and the grid entries) included
via create_graph=True. `L` is the scalar whose gradient NerfNetwork::backward_impl computes for a given dL/d(output):
rows 0..2 reach the colour MLP (extract_rgb), row 3 the sdf (add_density_gradient), rows 4..6 / B (the Eikonal term,
add_positions_view_ekloss) and rows 8..10 (add_positions_view) the normal, row 7 the variance (nerf_network.h:272-373).
tests/test_oracle_backward_autograd_cpu.py compares the CPU oracle's gradients with these, which pins the oracle's
backward / double-backward code with a model that shares none of it."""
import numpy as np
import torch


def grid_index(p, res, size):
    """Entry index of integer lattice points p [N,3] (grid.h:113-148): dense x + y res + z res^2 while the running stride
    stays within the table, else the xor-prime hash; always reduced modulo the table size."""
    p = p.astype(np.uint64)
    stride, dense_ok = 1, True
    index = np.zeros(p.shape[0], dtype=np.uint64)
    for d in range(3):
        if stride > size:
            dense_ok = False
            break
        index = index + p[:, d] * np.uint64(stride)
        stride *= int(res)
    if not dense_ok or size < stride:
        a = (p[:, 0] * np.uint64(1)) & np.uint64(0xFFFFFFFF)
        b = (p[:, 1] * np.uint64(2654435761)) & np.uint64(0xFFFFFFFF)
        c = (p[:, 2] * np.uint64(805459861)) & np.uint64(0xFFFFFFFF)
        index = a ^ b ^ c
    return (index % np.uint64(size)).astype(np.int64)


def encode(x, tables, offsets, resolution, scale, n_live):
    """x [N,3] float64 (requires_grad), tables: list of [size_l, 2] float64 tensors -> features [N, 28]."""
    N = x.shape[0]
    feats = []
    for l in range(14):
        if l >= len(tables) or l >= n_live:
            feats.append(torch.zeros(N, 2, dtype=torch.float64))
            continue
        size = int(offsets[l + 1] - offsets[l])
        pos = x * float(scale[l]) + 0.5
        cell = torch.floor(pos.detach())
        w = pos - cell
        cell_np = cell.numpy().astype(np.int64)
        f = torch.zeros(N, 2, dtype=torch.float64)
        for corner in range(8):
            d = np.array([(corner >> k) & 1 for k in range(3)], dtype=np.int64)
            idx = torch.from_numpy(grid_index(cell_np + d[None, :], int(resolution[l]), size))
            wt = torch.ones(N, dtype=torch.float64)
            for k in range(3):
                wt = wt * (w[:, k] if d[k] else (1.0 - w[:, k]))
            f = f + wt[:, None] * tables[l][idx]
        feats.append(f)
    return torch.cat(feats, dim=1)


def split_params(p, layout, offsets, n_levels):
    """The flat parameter vector (nerf_network.h:539-583) as float64 leaf tensors."""
    def leaf(a, shape):
        return torch.tensor(np.asarray(a, dtype=np.float64).reshape(shape), requires_grad=True)
    s, r, g, v = layout["sdf"], layout["rgb"], layout["grid"], layout["variance"]
    P = dict(W0=leaf(p[s:s + 2048], (64, 32)), W1=leaf(p[s + 2048:s + 3072], (16, 64)),
             C0=leaf(p[r:r + 3072], (64, 48)), C1=leaf(p[r + 3072:r + 7168], (64, 64)), C2=leaf(p[r + 7168:r + 8192], (16, 64)),
             var=leaf(p[v:v + 1], (1,)))
    P["tables"] = [leaf(p[g + 2 * int(offsets[l]):g + 2 * int(offsets[l + 1])], (-1, 2)) for l in range(n_levels)]
    return P


def loss_and_gradients(p, layout, offsets, resolution, scale, n_levels, n_live, coords, dout, batch_size, sdf_bias=-0.1, second_order=True):
    """Returns (flat gradient vector shaped like p, dict of forward quantities). second_order=False treats n = d sdf / dx as
    a constant w.r.t. the parameters (what a backward pass WITHOUT the double-backward terms would compute)."""
    P = split_params(p, layout, offsets, n_levels)
    x = torch.tensor(np.asarray(coords, dtype=np.float64)[:, :3], requires_grad=True)
    d = torch.tensor(np.asarray(dout, dtype=np.float64))
    N = x.shape[0]
    feat = encode(x, P["tables"], offsets, resolution, scale, n_live)
    h = torch.cat([x - 0.5, feat, torch.zeros(N, 1, dtype=torch.float64)], dim=1)
    z1 = torch.relu(h @ P["W0"].T)
    y = z1 @ P["W1"].T
    sdf = y[:, 0] + sdf_bias
    (n,) = torch.autograd.grad(y[:, 0].sum(), x, create_graph=True)
    n_used = n if second_order else n.detach()
    c_in = torch.cat([y, torch.zeros(N, 16, dtype=torch.float64), x, n_used, torch.zeros(N, 10, dtype=torch.float64)], dim=1)
    h1 = torch.relu(c_in @ P["C0"].T)
    h2 = torch.relu(h1 @ P["C1"].T)
    r = h2 @ P["C2"].T
    L = (d[:, 0:3] * r[:, 0:3]).sum() + (d[:, 3] * sdf).sum() + ((d[:, 4:7] / float(batch_size) + d[:, 8:11]) * n_used).sum() + (d[:, 7] * P["var"][0]).sum()
    leaves = [P["W0"], P["W1"], P["C0"], P["C1"], P["C2"], P["var"]] + P["tables"]
    grads = torch.autograd.grad(L, leaves, allow_unused=True)
    out = np.zeros(len(p), dtype=np.float64)
    s, rr, g, v = layout["sdf"], layout["rgb"], layout["grid"], layout["variance"]

    def put(lo, t, ref):
        a = (torch.zeros_like(ref) if t is None else t).detach().numpy().ravel()
        out[lo:lo + a.size] = a
    put(s, grads[0], P["W0"]); put(s + 2048, grads[1], P["W1"])
    put(rr, grads[2], P["C0"]); put(rr + 3072, grads[3], P["C1"]); put(rr + 7168, grads[4], P["C2"])
    put(v, grads[5], P["var"])
    for l in range(n_levels):
        put(g + 2 * int(offsets[l]), grads[6 + l], P["tables"][l])
    fwd = dict(sdf=sdf.detach().numpy(), normal=n.detach().numpy(), rgb=r.detach().numpy(), z1=z1.detach().numpy())
    return out, fwd


def relu_margins(p, layout, offsets, resolution, scale, n_levels, n_live, coords, sdf_bias=-0.1):
    """min |pre-activation| over the three hidden layers, per sample: samples with a comfortable margin switch their ReLUs
    identically in half and in float64 arithmetic."""
    P = split_params(p, layout, offsets, n_levels)
    x = torch.tensor(np.asarray(coords, dtype=np.float64)[:, :3], requires_grad=True)
    N = x.shape[0]
    feat = encode(x, P["tables"], offsets, resolution, scale, n_live)
    h = torch.cat([x - 0.5, feat, torch.zeros(N, 1, dtype=torch.float64)], dim=1)
    a1 = h @ P["W0"].T
    y = torch.relu(a1) @ P["W1"].T
    (n,) = torch.autograd.grad(y[:, 0].sum(), x, create_graph=False, retain_graph=True)
    c_in = torch.cat([y, torch.zeros(N, 16, dtype=torch.float64), x, n, torch.zeros(N, 10, dtype=torch.float64)], dim=1)
    a2 = c_in @ P["C0"].T
    a3 = torch.relu(a2) @ P["C1"].T
    m = torch.minimum(torch.minimum(a1.abs().min(dim=1).values, a2.abs().min(dim=1).values), a3.abs().min(dim=1).values)
    return m.detach().numpy()
