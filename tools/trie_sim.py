"""CPU model of the trie round (round 0 of the suffix sort): count first, move once.
Validates bucket formation, depth semantics and the doubling that follows against a plain suffix sort."""
import sys, numpy as np
import os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import datagen

def trie_round(x, SMALLMAX, MERGEMAX, DMAX, keybytes_bits):
    n = len(x)
    xp = np.concatenate([x, np.zeros(32, np.uint8)])
    # nodes: dict id -> (depth, start); children counts by level
    node_of = x.astype(np.int64).copy()          # depth-1 node = first byte
    depth_of_node = {v: 1 for v in range(256)}
    tot = np.bincount(x, minlength=256)
    node_start = dict(zip(range(256), np.concatenate([[0], np.cumsum(tot)[:-1]])))
    active = np.ones(n, bool)                    # suffix not yet in a leaf
    leaf_kind = np.zeros(n, np.int8)             # 1 small (bucket), 3 terminal
    leaf_val = np.zeros(n, np.int64)             # bucket id / head slot
    buckets = []                                 # (start, count, skip)
    next_node = 256
    level_nodes = list(range(256))
    stats = []
    for L in range(1, DMAX):
        idx = np.nonzero(active)[0]
        if len(idx) == 0: break
        dig = xp[idx + L].astype(np.int64)
        nd = node_of[idx]
        # counts per (node, digit)
        new_level = []
        order = np.lexsort((dig, nd))
        nds, dgs = nd[order], dig[order]
        keyc = nds * 256 + dgs
        uk, first, cnts = np.unique(keyc, return_index=True, return_counts=True)
        info = {}
        # per node sequential greedy
        pos = 0
        cur_node = -1
        for k, c in zip(uk, cnts):
            N, d = divmod(int(k), 256)
            if N != cur_node:
                # close bucket
                cur_node = N; run = node_start[N]; curb = -1
            start = run; run += int(c)
            if c > SMALLMAX:
                curb = -1
                if L + 1 < DMAX:
                    M = next_node; next_node += 1
                    depth_of_node[M] = L + 1; node_start[M] = start
                    info[k] = (2, M); new_level.append(M)
                else:
                    info[k] = (3, start)
            elif c > MERGEMAX:
                buckets.append([start, int(c), L]); info[k] = (1, len(buckets) - 1); curb = -1
            else:
                if curb >= 0 and buckets[curb][1] + c <= SMALLMAX:
                    buckets[curb][1] += int(c)
                else:
                    buckets.append([start, int(c), L]); curb = len(buckets) - 1
                info[k] = (1, curb)
        # apply to suffixes
        kinds = np.array([info[int(k)][0] for k in uk]); vals = np.array([info[int(k)][1] for k in uk])
        inv = np.searchsorted(uk, nd * 256 + dig)
        kk, vv = kinds[inv], vals[inv]
        settle = kk != 2
        leaf_kind[idx[settle]] = kk[settle]; leaf_val[idx[settle]] = vv[settle]
        node_of[idx[~settle]] = vv[~settle]
        active[idx[settle]] = False
        stats.append((L, len(idx) / n, len(new_level)))
        level_nodes = new_level
    assert not active.any()
    # scatter + bucket sort
    rank = np.zeros(n, np.int64); live = np.zeros(n, bool); sa = -np.ones(n, np.int64)
    t = leaf_kind == 3
    rank[t] = leaf_val[t]; live[t] = True
    barr = np.array(buckets, dtype=np.int64).reshape(-1, 3)
    sm = np.nonzero(leaf_kind == 1)[0]
    b_of = leaf_val[sm]
    skip = barr[b_of, 2]
    # key = keybits bits starting at byte skip
    nb = (keybytes_bits + 7) // 8
    key = np.zeros(len(sm), np.uint64)
    for j in range(nb):
        key = (key << np.uint64(8)) | xp[np.minimum(sm + skip + j, n + 31)].astype(np.uint64) * (sm + skip + j < n + 32)
    key >>= np.uint64(nb * 8 - keybytes_bits)
    order = np.lexsort((key, b_of))
    sm_s, key_s, b_s = sm[order], key[order], b_of[order]
    head = np.ones(len(sm_s), bool)
    head[1:] = (key_s[1:] != key_s[:-1]) | (b_s[1:] != b_s[:-1])
    # slot of element = bucket start + index within bucket
    bfirst = np.ones(len(sm_s), bool); bfirst[1:] = b_s[1:] != b_s[:-1]
    pos_in_all = np.arange(len(sm_s))
    bstart_pos = np.maximum.accumulate(np.where(bfirst, pos_in_all, 0))
    slot = barr[b_s, 0] + (pos_in_all - bstart_pos)
    headslot = slot[np.maximum.accumulate(np.where(head, pos_in_all, 0))]
    gid = np.cumsum(head) - 1
    gsz = np.bincount(gid)[gid]
    rank[sm_s] = headslot; live[sm_s] = gsz > 1
    fin = gsz == 1
    sa[slot[fin]] = sm_s[fin]
    return rank, live, sa, dict(levels=stats, nodes=next_node, buckets=len(buckets), terminal=t.mean(), bsizes=barr[:, 1])

def doubling(x, rank, live, sa, h):
    n = len(x)
    rounds = 0
    while live.any():
        s = np.nonzero(live)[0]
        j = s + h
        hcap = min(h, n)
        r2 = np.where(j < n, rank[np.minimum(j, n - 1)] + hcap + 1, n - s)
        g = rank[s]
        order = np.lexsort((r2, g))
        s_s, g_s, r_s = s[order], g[order], r2[order]
        head = np.ones(len(s_s), bool); head[1:] = (g_s[1:] != g_s[:-1]) | (r_s[1:] != r_s[:-1])
        seg = np.ones(len(s_s), bool); seg[1:] = g_s[1:] != g_s[:-1]
        p = np.arange(len(s_s))
        segpos = np.maximum.accumulate(np.where(seg, p, 0))
        slot = g_s + (p - segpos)
        headslot = np.maximum.accumulate(np.where(head, slot, 0))
        gid = np.cumsum(head) - 1
        gsz = np.bincount(gid)[gid]
        rank[s_s] = headslot; live[s_s] = gsz > 1
        fin = gsz == 1; sa[slot[fin]] = s_s[fin]
        h *= 2; rounds += 1
        print("   round h=%d live %.4f" % (h, live.mean()))
    return sa, rounds

def true_sa(x):
    n = len(x)
    b = x.tobytes()
    return np.array(sorted(range(n), key=lambda i: b[i:]), dtype=np.int64)

def check(trials=40, quiet=True):
    """the trie round (three settings of bucket capacity / merge limit / depth / key bits) + doubling from h = 6 reproduce the
    suffix array of random, tiny-alphabet, sparse, periodic, text-like and all-zero inputs"""
    import contextlib, io
    rng = np.random.default_rng(1)
    with (contextlib.redirect_stdout(io.StringIO()) if quiet else contextlib.nullcontext()):
        for trial in range(trials):
            n = int(rng.integers(1, 3000))
            kind = trial % 6
            if kind == 0: x = rng.integers(0, 256, n, dtype=np.uint8)
            elif kind == 1: x = rng.integers(0, 3, n, dtype=np.uint8)
            elif kind == 2: x = np.zeros(n, np.uint8); x[rng.integers(0, n, max(1, n // 10))] = 7
            elif kind == 3: x = np.tile(rng.integers(0, 256, 13, dtype=np.uint8), n // 13 + 1)[:n].copy()
            elif kind == 4: x = datagen.block(trial, n, 0)
            else: x = np.zeros(n, np.uint8)
            for (SM, MM, DM, KB) in ((15, 8, 7, 42), (63, 32, 6, 40), (5, 2, 7, 42)):
                rank, live, sa, st = trie_round(x, SM, MM, DM, KB)
                sa, _ = doubling(x, rank, live, sa, 6)
                assert (sa == true_sa(x)).all(), (trial, n, kind)
    return True


if __name__ == '__main__':
    mode = sys.argv[1] if len(sys.argv) > 1 else 'check'
    if mode == 'check':
        check(quiet=False)
        print("ok")
    else:
        n = 4 << 20
        for cls in range(5):
            x = datagen.block(cls, n, cls)
            for DM in (6, 7):
                rank, live, sa, st = trie_round(x, 8191, 4096, DM, 42)
                bs = st['bsizes']
                print("class %d Dmax %d: nodes %d buckets %d (mean %.0f, <1024: %d) terminal %.3f live %.3f levels %s" % (cls, DM, st['nodes'], st['buckets'], bs.mean(), (bs < 1024).sum(), st['terminal'], live.mean(), [(l, round(a, 3), k) for l, a, k in st['levels']]))
