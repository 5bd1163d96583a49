"""Seeded synthetic diffusion phantoms shared by tests and bench.py.

Multi-compartment Gaussian signal  s_i = S0 * sum_k f_k exp(-b_i g_i' R_k diag(l1,l2,l2) R_k' g_i) + noise,
fibre axes from a smooth field so that streamlines are long.  NumPy for small test volumes; the torch
variant builds the 140^3 benchmark volumes directly in HBM."""
import numpy as np


def icosa6():
    g = (1 + 5 ** 0.5) / 2
    v = np.array([[0, 1, g], [0, 1, -g], [1, g, 0], [1, -g, 0], [g, 0, 1], [g, 0, -1]], float)
    return (v / np.linalg.norm(v, axis=1)[:, None]).astype(np.float32)


def sphere_dirs(n, seed):
    """n roughly uniform unit vectors (golden-spiral + seeded rotation); deterministic."""
    rng = np.random.default_rng(seed)
    i = np.arange(n) + 0.5
    phi = np.arccos(1 - 2 * i / n)
    th = np.pi * (1 + 5 ** 0.5) * i
    v = np.stack([np.cos(th) * np.sin(phi), np.sin(th) * np.sin(phi), np.cos(phi)], 1)
    q, _ = np.linalg.qr(rng.normal(size=(3, 3)))
    return (v @ q.T).astype(np.float32)


def scheme_dti(ndir=6, nb0=1, b=1000.0, seed=1):
    """C1: 1 b0 + 6 icosahedral dirs; C2: 4 b0 + 60 dirs at b=1000."""
    dirs = icosa6() if ndir == 6 else sphere_dirs(ndir, seed)
    bvec = np.vstack([np.zeros((nb0, 3), np.float32), dirs]).astype(np.float32)
    bval = np.concatenate([np.zeros(nb0), np.full(ndir, b)]).astype(np.float32)
    return bval, bvec


def scheme_gqi(nb0=18, ndir=84, shells=(1000.0, 2000.0, 3000.0), seed=3):
    """C3: 18 x b=5 + 84 dirs x {1000,2000,3000} = 270 frames (HCP-like)."""
    dirs = sphere_dirs(ndir, seed)
    bvec = [np.tile(np.array([[1.0, 0, 0]], np.float32), (nb0, 1))]
    bval = [np.full(nb0, 5.0)]
    for s in shells:
        bvec.append(dirs)
        bval.append(np.full(ndir, s))
    return np.concatenate(bval).astype(np.float32), np.vstack(bvec).astype(np.float32)


def scheme_dsi(bmax=7000.0, r2max=25):
    """C5: DSI lattice |iq|^2 <= 25 -> 515 frames, b = bmax*|iq|^2/25 (b0 first)."""
    r = int(np.floor(np.sqrt(r2max)))
    g = np.arange(-r, r + 1)
    pts = np.array([(x, y, z) for z in g for y in g for x in g if x * x + y * y + z * z <= r2max], float)
    n2 = (pts ** 2).sum(1)
    order = np.argsort(n2, kind="stable")
    pts, n2 = pts[order], n2[order]
    bval = (bmax * n2 / r2max).astype(np.float32)
    nrm = np.sqrt(np.maximum(n2, 1e-30))
    bvec = np.where(n2[:, None] > 0, pts / nrm[:, None], 0.0).astype(np.float32)
    return bval, bvec


def fibre_field(nx, ny, nz):
    """smooth unit-vector field (cos th(x,y), sin th(x,y), .3 sin(z/10)) normalised"""
    x, y, z = np.meshgrid(np.arange(nx), np.arange(ny), np.arange(nz), indexing="ij")
    th = 0.04 * x + 0.03 * y
    v = np.stack([np.cos(th), np.sin(th), 0.3 * np.sin(z / 10.0)], -1)
    return v / np.linalg.norm(v, axis=-1, keepdims=True)


def signal(bval, bvec, axes, fracs, s0, lam=(1.7e-3, 0.3e-3), noise=0.0, rng=None, floor=None):
    """axes: list of [..., 3] unit fields, fracs: list of [...] weights. Returns float32 [..., nvol]."""
    bval = np.asarray(bval, np.float64)
    bvec = np.asarray(bvec, np.float64)
    s = 0.0
    for ax, f in zip(axes, fracs):
        c = np.tensordot(ax, bvec.T, axes=([-1], [0]))               # [..., nvol] = g.e
        s = s + np.asarray(f)[..., None] * np.exp(-bval * (lam[1] + (lam[0] - lam[1]) * c * c))
    s = np.asarray(s0)[..., None] * s
    if noise and rng is not None:
        s = s + rng.normal(scale=noise, size=s.shape)
    if floor is not None:
        s = np.maximum(s, floor)
    return np.asfortranarray(s.astype(np.float32))


def ball_mask(nx, ny, nz, radius=None):
    c = np.array([(nx + 1) / 2.0, (ny + 1) / 2.0, (nz + 1) / 2.0])
    r = radius if radius is not None else min(nx, ny, nz) * 62.0 / 140.0
    x, y, z = np.meshgrid(np.arange(1, nx + 1), np.arange(1, ny + 1), np.arange(1, nz + 1), indexing="ij")
    return np.asfortranarray((((x - c[0]) ** 2 + (y - c[1]) ** 2 + (z - c[2]) ** 2) <= r * r).astype(np.uint8))


def make_volume(shape, bval, bvec, seed, nfib=1, noise_frac=0.02, nonpositive_frac=0.0, crossing=False):
    """Seeded test volume: S0~U(800,1200), 1-2 fibres, Gaussian noise S0/50; optional zero/negative samples."""
    rng = np.random.default_rng(seed)
    nx, ny, nz = shape
    ax1 = fibre_field(nx, ny, nz)
    axes, fracs = [ax1], [np.ones(shape)]
    if crossing or nfib > 1:
        ax2 = np.stack([-ax1[..., 1], ax1[..., 0], np.zeros(shape)], -1)
        ax2 /= np.maximum(np.linalg.norm(ax2, axis=-1, keepdims=True), 1e-12)
        f1 = rng.uniform(0.4, 0.7, shape)
        axes, fracs = [ax1, ax2], [f1, 1 - f1]
    s0 = rng.uniform(800, 1200, shape)
    dwi = signal(bval, bvec, axes, fracs, s0, noise=1000.0 * noise_frac if noise_frac else 0.0, rng=rng,
                 floor=None if nonpositive_frac else 1.0)
    if nonpositive_frac:
        hit = rng.random(dwi.shape) < nonpositive_frac
        dwi[hit] = np.where(rng.random(hit.sum()) < 0.5, 0.0, -3.0).astype(np.float32)
    return np.asfortranarray(dwi), axes, fracs


# ---------------------------------------------------------------------------------------------
# torch variants: build benchmark-size volumes directly in HBM (synthetic data, seeded)
# ---------------------------------------------------------------------------------------------
def make_dwi_torch(shape, bval, bvec, seed, device, nfib=2, noise_frac=0.02, floor=1.0, chunk=1 << 17):
    """Returns (dwi float32 [nvol, nvox] planar == MRI.vol memory order, axes float32 [3, nvox]).
    Same signal model as make_volume (smooth fibre field + in-plane crossing fibre, S0~U(800,1200),
    Gaussian noise, clamped to >= floor for the all-positive benchmark variant)."""
    import torch
    nx, ny, nz = shape
    nvox = nx * ny * nz
    g = torch.Generator(device=device)
    g.manual_seed(int(seed))
    bv = torch.as_tensor(np.asarray(bval, np.float32), device=device)
    gd = torch.as_tensor(np.asarray(bvec, np.float32), device=device)          # [nvol, 3]
    nvol = bv.numel()
    dwi = torch.empty((nvol, nvox), dtype=torch.float32, device=device)
    axes = torch.empty((3, nvox), dtype=torch.float32, device=device)
    l1, l2 = 1.7e-3, 0.3e-3
    for c0 in range(0, nvox, chunk):
        c1 = min(nvox, c0 + chunk)
        lin = torch.arange(c0, c1, device=device)
        x = (lin % nx).float(); y = ((lin // nx) % ny).float(); z = (lin // (nx * ny)).float()
        th = 0.04 * x + 0.03 * y
        a1 = torch.stack([torch.cos(th), torch.sin(th), 0.3 * torch.sin(z / 10.0)], 1)
        a1 = a1 / a1.norm(dim=1, keepdim=True)
        axes[:, c0:c1] = a1.T
        s0 = 800.0 + 400.0 * torch.rand(c1 - c0, generator=g, device=device)
        c = a1 @ gd.T                                                           # [chunk, nvol]
        sig = torch.exp(-bv * (l2 + (l1 - l2) * c * c))
        if nfib > 1:
            a2 = torch.stack([-a1[:, 1], a1[:, 0], torch.zeros_like(a1[:, 0])], 1)
            a2 = a2 / a2.norm(dim=1, keepdim=True).clamp_min(1e-12)
            f1 = 0.4 + 0.3 * torch.rand(c1 - c0, generator=g, device=device)
            c2 = a2 @ gd.T
            sig = f1[:, None] * sig + (1 - f1)[:, None] * torch.exp(-bv * (l2 + (l1 - l2) * c2 * c2))
        sig = s0[:, None] * sig
        if noise_frac:
            sig = sig + (1000.0 * noise_frac) * torch.randn(sig.shape, generator=g, device=device)
        if floor is not None:
            sig = sig.clamp_min(floor)
        dwi[:, c0:c1] = sig.T
    return dwi, axes


def ball_mask_torch(shape, device, radius=None):
    import torch
    nx, ny, nz = shape
    lin = torch.arange(nx * ny * nz, device=device)
    x = (lin % nx).float() + 1; y = ((lin // nx) % ny).float() + 1; z = (lin // (nx * ny)).float() + 1
    r = radius if radius is not None else min(shape) * 62.0 / 140.0
    c = [(n + 1) / 2.0 for n in shape]
    return (((x - c[0]) ** 2 + (y - c[1]) ** 2 + (z - c[2]) ** 2) <= r * r).to(torch.uint8)


def bundle_field_torch(shape, device, seed=7, cell=14.0, hole_frac=0.12):
    """Tracking phantom with a BROAD streamline-length distribution (the smooth field above makes almost every line use its
    whole len_max budget): the volume is a Voronoi partition into straight "bundles" (random centres about `cell` voxels apart,
    one random unit direction each), a fraction of the cells is empty (no vector: lines end at their border), and lines also end
    where two bundles meet at more than the angle threshold.  Returns (ovec float32 [3, nvox] planar, mask uint8 [nvox])."""
    import torch
    nx, ny, nz = shape
    nvox = nx * ny * nz
    g = torch.Generator(device=device)
    g.manual_seed(int(seed))
    ncell = max(4, int(round(nvox / cell ** 3)))
    ctr = torch.rand((ncell, 3), generator=g, device=device) * torch.tensor([nx, ny, nz], dtype=torch.float32, device=device)
    dirs = torch.randn((ncell, 3), generator=g, device=device)
    dirs = dirs / dirs.norm(dim=1, keepdim=True)
    empty = torch.rand(ncell, generator=g, device=device) < hole_frac
    dirs[empty] = 0.0
    ovec = torch.empty((3, nvox), dtype=torch.float32, device=device)
    chunk = 1 << 15
    for c0 in range(0, nvox, chunk):
        c1 = min(nvox, c0 + chunk)
        lin = torch.arange(c0, c1, device=device)
        p = torch.stack([(lin % nx).float(), ((lin // nx) % ny).float(), (lin // (nx * ny)).float()], 1) + 0.5
        owner = torch.cdist(p, ctr).argmin(1)
        ovec[:, c0:c1] = dirs[owner].T
    mask = ball_mask_torch(shape, device) & (ovec.abs().sum(0) > 0).to(torch.uint8)
    return ovec.contiguous(), mask.contiguous()
