#!/usr/bin/env python
"""The one check this repository cannot run itself (VERDICT r4 missing #6): does a REAL MinkowskiEngine
0.5.4 number its kernel offsets the way the oracle and the HIP kernels do?  With seeded random weights any self-consistent
convention passes parity; with TRAINED weights (diff_net.ckpt / refine_net.ckpt) the k <-> offset convention decides the result.

    python tools/me_convention_check.py              on a box with MinkowskiEngine==0.5.4 (CPU build is enough)

For each of LiDiff's three convolution kinds -- kernel_size 3 / stride 1, kernel_size 2 / stride 2, transposed kernel_size 2 /
stride 2 (minkunet.py:53-66, 13-29, 32-46) -- and every kernel index k: a one-hot kernel (kernel[k, 0, 0] = 1) over a full 3x3x3
(resp. 4x4x4) block of voxels with distinct features; the output rows tell which input voxel ME read for offset k.  The same
computation runs on this repository's oracle (always) and, if a GPU is present, on lidiff_amd's ME shim; the script prints the
k -> (dx, dy, dz) table of each implementation and FAILS if they differ.  Without MinkowskiEngine it prints the oracle's table
(the convention this repository assumes: x fastest, odd kernels centred, even kernels {0, +ts}) and exits 0 with a notice."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def block(n, step=1):
    g = np.arange(n) * step
    xs, ys, zs = np.meshgrid(g, g, g, indexing="ij")
    c = np.stack([np.zeros(n ** 3, np.int64), xs.ravel(), ys.ravel(), zs.ravel()], 1).astype(np.int32)
    feats = (1.0 + np.arange(c.shape[0], dtype=np.float32))[:, None]          # distinct: the value names the input voxel
    return c, feats


def offsets_from_outputs(in_c, in_f, out_c, out_f):
    """For every output row with a non-zero value: the input voxel it read (by its feature value) minus the output voxel."""
    votes = {}
    for o in np.nonzero(np.abs(out_f[:, 0]) > 0.5)[0]:
        src = int(round(float(out_f[o, 0]))) - 1
        d = tuple(int(v) for v in (in_c[src, 1:] - out_c[o, 1:]))
        votes[d] = votes.get(d, 0) + 1
    return max(votes, key=votes.get) if votes else None


def run_oracle(kind):
    from oracle import me_cpu as me
    table = {}
    if kind == "k3s1":
        c, f = block(3)
        uniq, inv, _ = me.voxelize(c)
        nbr = me.kernel_map(uniq, uniq, 3, 1)
        for k in range(27):
            w = torch.zeros(27, 1, 1)
            w[k] = 1
            out = me.conv_forward(torch.from_numpy(f[np.argsort(inv)]), w, nbr).numpy()
            table[k] = offsets_from_outputs(uniq, f[np.argsort(inv)], uniq, out)
    else:
        c, f = block(4)
        fine, inv, _ = me.voxelize(c)
        ff = f[np.argsort(inv)]
        coarse, _, _ = me.voxelize(me.floor_to_stride(fine, 2))
        down = me.kernel_map(fine, coarse, 2, 1)
        if kind == "k2s2":
            for k in range(8):
                w = torch.zeros(8, 1, 1)
                w[k] = 1
                table[k] = offsets_from_outputs(fine, ff, coarse, me.conv_forward(torch.from_numpy(ff), w, down).numpy())
        else:
            up = me.transpose_kernel_map(down, fine.shape[0])
            cf = (1.0 + np.arange(coarse.shape[0], dtype=np.float32))[:, None]
            for k in range(8):
                w = torch.zeros(8, 1, 1)
                w[k] = 1
                table[k] = offsets_from_outputs(coarse, cf, fine, me.conv_forward(torch.from_numpy(cf), w, up).numpy())
    return table


CO = 16        # output channels of the one-hot convolutions (every channel carries the same value; lidiff_amd's kernels want 16 | C_out)


def run_me_api(ME, kind, device):
    """The same through an ME-compatible module tree (the real MinkowskiEngine, or lidiff_amd's shim on a GPU)."""
    table = {}
    n = 3 if kind == "k3s1" else 4
    c, f = block(n)
    coords, feats = torch.from_numpy(c).to(device), torch.from_numpy(f).to(device)
    for k in range(27 if kind == "k3s1" else 8):
        x = ME.SparseTensor(features=feats, coordinates=coords, device=device)
        if kind == "k3s1":
            conv = ME.MinkowskiConvolution(1, CO, kernel_size=3, stride=1, dimension=3).to(device)
        else:
            conv = ME.MinkowskiConvolution(1, CO, kernel_size=2, stride=2, dimension=3).to(device)
        with torch.no_grad():
            conv.kernel.zero_()
            conv.kernel[k, 0, :] = 1.0
            y = conv(x)
            if kind == "k2s2T":
                # feed the coarse map back up through the transposed convolution: one-hot at k there, all-ones below
                conv.kernel.fill_(0.0)
                conv.kernel[:, 0, :] = 1.0
                y = conv(x)
                up = ME.MinkowskiConvolutionTranspose(1, CO, kernel_size=2, stride=2, dimension=3).to(device)
                up.kernel.zero_()
                up.kernel[k, 0, :] = 1.0
                cf = (1.0 + torch.arange(y.F.shape[0], dtype=torch.float32, device=device))[:, None]
                if hasattr(y, "coordinate_map_key"):          # the real ME
                    y_in = ME.SparseTensor(features=cf, coordinate_map_key=y.coordinate_map_key, coordinate_manager=y.coordinate_manager)
                else:                                          # lidiff_amd's shim
                    y_in = type(y)(cf, tensor_stride=y.tensor_stride, coordinate_manager=y.coordinate_manager)
                z = up(y_in)
                table[k] = offsets_from_outputs(y.C.cpu().numpy(), cf.cpu().numpy(), z.C.cpu().numpy(), z.F.cpu().numpy())
                continue
        in_c = x.C.cpu().numpy()
        in_f = x.F.cpu().numpy()
        table[k] = offsets_from_outputs(in_c, in_f, y.C.cpu().numpy(), y.F.cpu().numpy())
    return table


def main():
    tables = {}
    for kind in ("k3s1", "k2s2", "k2s2T"):
        tables[("oracle", kind)] = run_oracle(kind)
    if torch.cuda.is_available():
        import lidiff_amd.MinkowskiEngine as LME
        for kind in ("k3s1", "k2s2", "k2s2T"):
            tables[("lidiff_amd", kind)] = run_me_api(LME, kind, torch.device("cuda:0"))
    try:
        import MinkowskiEngine as RME
        if not hasattr(RME, "__version__") or "lidiff_amd" in getattr(RME, "__file__", ""):
            raise ImportError("the MinkowskiEngine on this path is lidiff_amd's alias")
        dev = torch.device("cuda:0" if torch.cuda.is_available() else "cpu")
        for kind in ("k3s1", "k2s2", "k2s2T"):
            tables[("MinkowskiEngine " + RME.__version__, kind)] = run_me_api(RME, kind, dev)
    except ImportError as e:
        print(f"NOTE: no real MinkowskiEngine importable here ({e}); printing this repository's convention only.")
    bad = 0
    for kind in ("k3s1", "k2s2", "k2s2T"):
        ref = tables[("oracle", kind)]
        print(f"--- {kind}: k -> (input voxel - output voxel), oracle: " + " ".join(f"{k}:{ref[k]}" for k in sorted(ref)))
        for (impl, kd), tab in tables.items():
            if kd != kind or impl == "oracle":
                continue
            same = tab == ref
            bad += 0 if same else 1
            print(f"    {impl}: {'SAME' if same else 'DIFFERENT: ' + ' '.join(f'{k}:{tab[k]}' for k in sorted(tab))}")
    # the convention in words (Appendix A.6)
    ref = tables[("oracle", "k3s1")]
    assert all(ref[k] == ((k % 3) - 1, (k // 3) % 3 - 1, k // 9 - 1) for k in range(27)), "oracle: k3 is not x-fastest / centred"
    ref = tables[("oracle", "k2s2")]
    assert all(ref[k] == (k % 2, (k // 2) % 2, k // 4) for k in range(8)), "oracle: k2 is not x-fastest / {0, +ts}"
    print("convention assumed here: kernel_size 3: k = (dx+1) + 3 (dy+1) + 9 (dz+1); kernel_size 2: k = dx + 2 dy + 4 dz, input = "
          "output + offset * tensor_stride; the transposed map is the swap of the strided one at the same k")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
