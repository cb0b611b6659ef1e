# Brute-force a 16-byte-chunk XOR swizzle for an UNPADDED [rows][64] bf16 LDS tile (128-byte rows) that is conflict-free for
#  (a) the row-fragment reads (ds_read_b128; lane (l15, lg): row = r0 + l15, chunk = 4 ks + lg) and
#  (b) the transposed column-fragment reads (two ds_read_b64_tr_b16; lane (l15, lg): row = r0 + 4 lg + (l15 >> 2) [+16], byte = 32 dt + 8 (l15 & 3))
# under the bank model of MI355X_MICROARCH.md (b128: four groups of 16 lanes, distinct 16-byte slots of a 256-byte line; b64: two groups of 32
# lanes, distinct 8-byte slots).
import itertools
B128_GROUPS = [[0,1,2,3,12,13,14,15,20,21,22,23,24,25,26,27], [4,5,6,7,8,9,10,11,16,17,18,19,28,29,30,31],
               [32,33,34,35,44,45,46,47,52,53,54,55,56,57,58,59], [36,37,38,39,40,41,42,43,48,49,50,51,60,61,62,63]]
def addr(row, byte, f):            # byte offset inside the tile after swizzling the 16-byte chunk index
    c, within = byte // 16, byte % 16
    return row * 128 + ((c ^ f(row)) * 16) + within
def ok(f):
    for ks in range(2):
        for grp in B128_GROUPS:
            slots = set()
            for lane in grp:
                l15, lg = lane & 15, lane >> 4
                a = addr(l15, (4 * ks + lg) * 16, f)
                s = (a // 16) % 16
                if s in slots: return False
                slots.add(s)
    for dt in range(4):
        for half in range(2):          # second read: rows + 16
            for grp in (range(0, 32), range(32, 64)):
                slots = set()
                for lane in grp:
                    l15, lg = lane & 15, lane >> 4
                    row = 4 * lg + (l15 >> 2) + 16 * half
                    a = addr(row, 32 * dt + 8 * (l15 & 3), f)
                    s = (a // 8) % 32
                    if s in slots: return False
                    slots.add(s)
    return True
found = []
for m in range(1 << 15):              # f(row) = linear in the five row bits: three output bits x five input bits
    rowsbits = [(m >> (5 * o)) & 31 for o in range(3)]
    def f(row, rb=rowsbits):
        r = row & 31
        return sum(((bin(r & rb[o]).count('1') & 1) << o) for o in range(3))
    if ok(f):
        found.append(rowsbits)
print(len(found), found[:10])
# model check: the padded layout in use (160-byte rows, no swizzle) must come out conflict-free, the unpadded unswizzled one must not
def ok_stride(stride, f):
    def ad(row, byte): return row * stride + (((byte // 16) ^ f(row)) * 16) + byte % 16
    for ks in range(2):
        for grp in B128_GROUPS:
            s = [(ad(l & 15, (4 * ks + (l >> 4)) * 16) // 16) % 16 for l in grp]
            if len(set(s)) < 16: return 'b128 conflict'
    for dt in range(4):
        for half in range(2):
            for grp in (range(0, 32), range(32, 64)):
                s = [(ad(4 * (l >> 4) + ((l & 15) >> 2) + 16 * half, 32 * dt + 8 * (l & 3)) // 8) % 32 for l in grp]
                if len(set(s)) < 32: return 'tr conflict'
    return 'conflict-free'
print('160-byte rows, no swizzle:', ok_stride(160, lambda r: 0))
print('128-byte rows, no swizzle:', ok_stride(128, lambda r: 0))
f = lambda r: (((r >> 2) & 1) << 1) | (((r >> 1) & 1) << 2)
print('128-byte rows, f = [0,4,2]:', ok_stride(128, f))
