"""What would starting the heaviest tiles first buy?  List-scheduling replay of one tick's pass A / pass B from the
per-wave timeline (scripts/pile_timeline.py ... out.npz): every XCD runs its own blocks in index order on
`slots` workgroup slots; a block lives as long as it did in the measured tick.  Orders: as measured (the model's
baseline), heaviest first within each XCD's run (perfect knowledge), and heaviest first by the weights of ANOTHER tick
(second file: the previous tick's durations as the predictor).
   python scripts/tile_order_sim.py tick.npz [previous_tick.npz]"""
import sys
import heapq
import numpy as np

def blocks(st):
    w = st.reshape(-1, 4, 4)  # four waves per workgroup
    ok = (w[:, :, 0] > 0).all(axis=1)
    start = np.where(ok, w[:, :, 0].min(axis=1), 0); end = np.where(ok, w[:, :, 1].max(axis=1), 0)
    return ok, (end - start) * 0.01, (start - start[ok].min()) * 0.01, (end - start[ok].min()) * 0.01

def replay(dur, order_of_xcd, slots):
    finish = 0.0
    for x in range(8):
        free = [0.0] * slots
        heapq.heapify(free)
        for b in order_of_xcd[x]:
            t0 = heapq.heappop(free)
            heapq.heappush(free, t0 + dur[b])
            finish = max(finish, t0 + dur[b])
    return finish

cur = np.load(sys.argv[1]); prev = np.load(sys.argv[2]) if len(sys.argv) > 2 else None
for key, label, slots in (("a", "pass A", 160), ("b", "pass B", 128)):
    ok, dur, s, e = blocks(cur[key])
    nb = int(ok.sum())
    ids = [np.array([b for b in range(nb) if b % 8 == x]) for x in range(8)]
    base = replay(dur, ids, slots)
    best = replay(dur, [i[np.argsort(-dur[i], kind="stable")] for i in ids], slots)
    line = f"{label}: measured span {e[ok].max():6.1f} us; replay in index order {base:6.1f}; heaviest first (perfect) {best:6.1f}; sum of lives / slots {dur[ok].sum() / (8 * slots):6.1f}"
    if prev is not None:
        _, pd, _, _ = blocks(prev[key])
        # the previous tick's block b worked on tile tile_of(b); with the same mapping block b of this tick has the same tile
        line += f"; heaviest first by the previous tick's lives {replay(dur, [i[np.argsort(-pd[i], kind='stable')] for i in ids], slots):6.1f}"
        line += f" (in 8 classes of 2.56 us: {replay(dur, [i[np.argsort(-np.minimum(pd[i] // 2.56, 31), kind='stable')] for i in ids], slots):6.1f})"
    print(line)

# pass B ordered by what pass A of the SAME tick measured for the tile (pass A: ends-first mapping, pass B: plain runs)
def tile_plain(b, nb):
    q, r, x = nb >> 3, nb & 7, b & 7
    return x * q + min(x, r) + (b >> 3)
def tile_ends_first(b, nb):
    q, r, x = nb >> 3, nb & 7, b & 7
    start, ln, l = x * q + min(x, r), q + (1 if x < r else 0), b >> 3
    return start + ln - 1 - (l >> 1) if l & 1 else start + (l >> 1)
oka, dura, _, _ = blocks(cur["a"]); okb, durb, _, eb = blocks(cur["b"])
nb = int(okb.sum())
life_a_of_tile = np.zeros(nb)
for b in range(nb):
    life_a_of_tile[tile_ends_first(b, nb)] = dura[b]
key = np.array([life_a_of_tile[tile_plain(b, nb)] for b in range(nb)])
ids = [np.array([b for b in range(nb) if b % 8 == x]) for x in range(8)]
print(f"pass B by pass A's life of the same tile, same tick: {replay(durb, [i[np.argsort(-key[i], kind='stable')] for i in ids], 128):6.1f} us"
      f" (32 classes of 2.56 us: {replay(durb, [i[np.argsort(-np.minimum(key[i] // 2.56, 31), kind='stable')] for i in ids], 128):6.1f});"
      f" correlation of the two lives {np.corrcoef(key, durb[:nb])[0, 1]:.2f}")
