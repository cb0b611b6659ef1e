"""Brute-force LDS bank-conflict check of the attention-backward tile image (csrc/attention.hip, attn_bwd_dma_kernel):
tile[row block of 16][column half of 32][16 rows][64 bytes], 16-byte chunk x of row r stored at x ^ g4(r >> 2), g4 = [0,2,3,1].
Banking rules from MI355X_MICROARCH.md (LDS): ds_read_b128 = four 16-lane groups, bank = (addr/4) % 64;
ds_read_b64_tr_b16 = two 32-lane halves, bank = (addr/4) % 64.  Prints the worst multiplicity per access pattern (1 = conflict-free)."""
G4 = [0, 2, 3, 1]
B128_GROUPS = [[0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27], [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31]]
B128_GROUPS += [[l + 32 for l in g] for g in B128_GROUPS]


def off(row, col):          # byte offset inside a tile
    rb, r, h, c = row >> 4, row & 15, col >> 5, col & 31
    lc, e = c >> 3, c & 7
    return ((rb * 2 + h) * 512 + r * 32 + ((lc ^ G4[r >> 2]) * 8) + e) * 2


def worst(groups, addr_of_lane, width):
    w = 1
    for g in groups:
        banks = {}
        for l in g:
            a = addr_of_lane(l)
            for k in range(width // 4):
                banks.setdefault(((a // 4) + k) % 64, set()).add(a // 4 + k)
        w = max(w, max(len(v) for v in banks.values()))
    return w


res = {}
for row0 in (0, 16, 32, 208):
    for ks in (0, 1):      # row_frag: lane (l15, lg) reads row row0 + l15, columns 32 ks + 8 lg .. +7
        res[f'row_frag row0={row0} ks={ks}'] = worst(B128_GROUPS, lambda l: off(row0 + (l & 15), 32 * ks + 8 * (l >> 4)), 16)
for r0 in (0, 32, 192):
    for dt in range(4):    # col_frag: lane (l15, lg) reads row r0 + 4 lg + (l15 >> 2) (+16), columns 16 dt + 4 (l15 & 3) .. +3
        for add in (0, 16):
            res[f'col_frag r0={r0} dt={dt} +{add}'] = worst([list(range(32)), list(range(32, 64))],
                                                            lambda l: off(r0 + add + 4 * (l >> 4) + ((l & 15) >> 2), 16 * dt + 4 * (l & 3)), 8)
# delta pass: thread (row = tid >> 1, half = tid & 1) reads chunks i = 0..3 of its half: b128 by 64 consecutive threads
for i in range(4):
    res[f'delta chunk {i}'] = worst(B128_GROUPS, lambda l: off(l >> 1, 32 * (l & 1) + 8 * i), 16)
bad = {k: v for k, v in res.items() if v > 1}
print('patterns checked:', len(res), ' worst multiplicity:', max(res.values()))
print('conflicting patterns:', bad if bad else 'none')
