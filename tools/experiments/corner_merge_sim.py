import sys, os, time, numpy as np, torch
sys.path.insert(0, '/root/repo')
from shine_mapping_amd import synth
from shine_mapping_amd.feature_octree import morton_encode, morton_decode, FeatureOctree
kind, frames, az, pts, L, tiles_per_wave = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]), int(sys.argv[6])
cache = "/tmp/sim_%s_%d_%d_%d_L%d.npz" % (kind, frames, az, pts, L)
if os.path.exists(cache):
    z = np.load(cache)
    keys = [z["keys%d" % s] for s in range(L)]; nks = [z["nk%d" % s] for s in range(L)]; nids = [z["nid%d" % s] for s in range(L)]
else:
    t0 = time.time()
    cfg = synth.make_config(kind, device="cpu", tree_level_feat=L)
    octree = FeatureOctree(cfg)
    coords = []
    for c, l, w in synth.make_frames(cfg, frames, 64, az, 42, "cpu"):
        octree.update(c[w > 0], False)
        coords.append(c)
    pool = torch.cat(coords)
    print("built %s in %.1f s: pool %d, rows %s" % (kind, time.time() - t0, pool.shape[0], octree._corner_count), flush=True)
    g = torch.Generator().manual_seed(1)
    idx = torch.randint(0, pool.shape[0], (pts,), generator=g)
    c = pool[idx].numpy()
    keys = []
    for s in range(L):
        lvl = octree.free_level_num + s
        res = np.float32(2 ** lvl)
        v = np.floor(np.clip(res * (c.astype(np.float32) + np.float32(1.0)) * np.float32(0.5), 0, res - 1)).astype(np.int64)
        keys.append(morton_encode(v))
    nks, nids = octree._node_keys, octree._node_ids
    np.savez(cache, **{"keys%d" % s: keys[s] for s in range(L)}, **{"nk%d" % s: nks[s] for s in range(L)}, **{"nid%d" % s: nids[s] for s in range(L)})
order = np.argsort(keys[L - 1], kind="stable")
W = tiles_per_wave * 16
OFF = np.array([[(c >> 2) & 1, (c >> 1) & 1, c & 1] for c in range(8)])
def count(rows, wave, slot, S):
    t = np.arange(rows.size)
    key2 = wave.astype(np.int64) * S + slot
    o2 = np.lexsort((t, key2))
    r2, k2 = rows[o2], key2[o2]
    newseg = np.ones(rows.size, bool)
    newseg[1:] = (k2[1:] != k2[:-1]) | (r2[1:] != r2[:-1])
    return int(newseg.sum())
res = {}
for s in range(L):
    k = keys[s][order]
    nk, nid = nks[s], nids[s]
    o = np.argsort(nk)
    pos = np.searchsorted(nk[o], k); pos[pos >= nk.size] = nk.size - 1
    at = o[pos]
    hit = nk[at] == k
    ids = np.where(hit[:, None], nid[at], -1)
    n = k.shape[0]
    first = np.ones(n, bool); first[1:] = k[1:] != k[:-1]; first[np.arange(0, n, W)] = True
    sel = first & hit
    run_rows = ids[sel]; run_wave = np.nonzero(sel)[0] // W
    v = morton_decode(k[sel])                       # [runs, 3] node voxel coords
    cc = v[:, None, :] + OFF[None, :, :]            # [runs, 8, 3] corner lattice coords
    rows = run_rows.reshape(-1); wave = np.repeat(run_wave, 8)
    res.setdefault("base", 0); res["base"] += rows.size
    for name, slot, S in (("hash64", rows % 64, 64), ("hash128", rows % 128, 128),
                          ("lat4x4x4", ((cc[..., 0] & 3) << 4 | (cc[..., 1] & 3) << 2 | (cc[..., 2] & 3)).reshape(-1), 64),
                          ("lat8x4x2", ((cc[..., 0] & 7) << 3 | (cc[..., 1] & 3) << 1 | (cc[..., 2] & 1)).reshape(-1), 64),
                          ("lat4x8x2", ((cc[..., 0] & 3) << 4 | (cc[..., 1] & 7) << 1 | (cc[..., 2] & 1)).reshape(-1), 64),
                          ("lat8x8x2", ((cc[..., 0] & 7) << 4 | (cc[..., 1] & 7) << 1 | (cc[..., 2] & 1)).reshape(-1), 128),
                          ("lat4x4x2", ((cc[..., 0] & 3) << 3 | (cc[..., 1] & 3) << 1 | (cc[..., 2] & 1)).reshape(-1), 32)):
        res[name] = res.get(name, 0) + count(rows, wave, slot, S)
    print("level %d: runs %d rows %d" % (s, run_rows.shape[0], rows.size), flush=True)
print(kind, "row-atomics base", res["base"], {k: "%d (x%.2f)" % (v, res["base"] / v) for k, v in res.items() if k != "base"})
