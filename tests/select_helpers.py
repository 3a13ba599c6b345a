"""The loop of reprojector_utils::matchCandidates (reprojector.cpp:342-382) over given matches, in Python, and random lists for it:
shared by the CPU test (oracle against this) and the GPU test (device against both)."""
import numpy as np


def sequential(cell, success, occupancy, max_n, n_features):
    occ = occupancy.copy()
    visited = np.zeros(cell.size, np.uint8)
    i = trials = matches = 0
    for k in range(cell.size):
        i += 1
        c = int(cell[k])
        if c < 0 or c >= occ.size:
            continue
        if max_n > 0 and occ[c]:
            continue
        trials += 1
        visited[k] = 1
        if success[k]:
            matches += 1
            n_features += 1
            occ[c] = 1
            if max_n > 0 and n_features >= max_n:
                break
    return visited, occ, trials, matches, i, n_features


def make_lists(rng, n_lists, n_cells):
    lists = []
    for l in range(n_lists):
        kind = l % 7
        n = int(rng.randint(0, 3000)) if kind else 0                      # an empty list among them
        cell = rng.randint(0, n_cells, n).astype(np.int32)
        if kind == 1:
            cell[::17] = -1; cell[5::29] = n_cells + 3                    # outside the grid: never tried
        p = (0.05, 0.3, 0.6, 0.0, 1.0, 0.4, 0.2)[kind]
        success = (rng.uniform(size=n) < p).astype(np.uint8)
        occ = (rng.uniform(size=n_cells) < (0.0, 0.3, 0.7, 0.2, 0.1, 1.0, 0.5)[kind]).astype(np.uint8)
        n_before = int(rng.randint(0, 200))
        max_n = int(rng.choice([1, 50, 180, 181, 400, 100000]))           # below, at and far above what the list can add
        lists.append((cell, success, occ, max_n, n_before))
    return lists
