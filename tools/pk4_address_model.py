#!/usr/bin/env python
"""Address model of vl_gemm_pk4.hip (no GPU): the LDS placement of the DMA (unit i of wave w = rows i*32 + w*8 + (lane>>3), 16-byte
chunk (lane&7) ^ ((row>>1)&7)) against the fragment reads of the MFMA loop (row = wave_m*128 + ia*16 + (lane&15), chunk
(h*4 + (lane>>4)) ^ ((row>>1)&7)), the bank footprint of one ds_read_b128, and the epilogue's accumulator-layout -> slab ->
store-layout round trip.  Written while the kernel could not be run; every mismatch count must be 0."""
lds = {}
for wid in range(4):
    for i in range(8):
        for l in range(64):
            drow = wid * 8 + (l >> 3)
            addr = (i * 4 + wid) * 1024 + l * 16
            assert addr not in lds
            lds[addr] = (i * 32 + drow, (l & 7) ^ ((drow >> 1) & 7))
assert len(lds) == 256 * 8
bad = 0
for wave_m in range(2):
    for ia in range(8):
        for h in range(2):
            for l in range(64):
                fr16, fq = l & 15, l >> 4
                addr = (wave_m * 128 + fr16) * 128 + ia * 2048 + (((h * 4 + fq) ^ ((fr16 >> 1) & 7)) * 16)
                bad += lds[addr] != (wave_m * 128 + ia * 16 + fr16, h * 4 + fq)
print("fragment reads that hit the wrong (row, k chunk):", bad)
worst = 0
for h in range(2):
    for grp in range(4):
        banks = {}
        for l in range(grp * 16, grp * 16 + 16):
            fr16, fq = l & 15, l >> 4
            addr = fr16 * 128 + (((h * 4 + fq) ^ ((fr16 >> 1) & 7)) * 16)
            for b in range(4):
                banks[((addr >> 2) + b) & 63] = banks.get(((addr >> 2) + b) & 63, 0) + 1
        worst = max(worst, max(banks.values()))
print("lanes per bank within a 16-lane group of ds_read_b128 (1 = conflict-free):", worst)
slab, bad2 = {}, 0
for l in range(64):
    er, eq = l & 15, l >> 4
    for jh in range(2):
        for q in range(4):
            row = jh * 16 + er
            base = row * 128 + (((q * 2 + (eq >> 1)) ^ (row & 7)) << 4) + (eq & 1) * 8
            for e in range(4):
                slab[base + e * 2] = (row, q * 16 + eq * 4 + e)
for l in range(64):
    prow, pchunk = l >> 3, l & 7
    for ps in range(4):
        r = ps * 8 + prow
        a = r * 128 + ((pchunk ^ (r & 7)) << 4)
        for e in range(8):
            bad2 += slab[a + e * 2] != (r, pchunk * 8 + e)
print("slab reads that hit the wrong (row, column):", bad2)
assert bad == 0 and worst == 1 and bad2 == 0


# ---- schedule model of the main loop: which (tile, k-step) a stage holds when it is read, and that no DMA batch overwrites
# data still to be read (two stages, DMA two k-steps ahead of the MFMAs and across tile boundaries) ----
def run_schedule(my_tiles, nk):
    stage = [None, None]
    unread = [set(), set()]                               # k halves of the held data the MFMA stream has not fetched yet
    pos = [0, 0]                                          # DMA position (tile, k-step)

    def dma(st):
        assert not unread[st], ("DMA overwrites unread data", st, stage[st], unread[st])
        stage[st] = tuple(pos); unread[st] = {0, 1}
        pos[1] += 1
        if pos[1] == nk:
            pos[1] = 0; pos[0] += 1

    def read(st, h, want):
        assert stage[st] == want and h in unread[st], ("wrong or repeated read", st, h, stage[st], want)
        unread[st].discard(h)
    dma(0); dma(1)
    par = 0
    read(0, 0, (0, 0))                                    # ldfrags(smem, 0, 0)
    for ti in range(my_tiles):
        for kt in range(nk):
            cur, oth = par, par ^ 1
            read(cur, 1, (ti, kt))                        # phase A: the fragments of half 1
            if pos[0] < my_tiles:                         # (barrier) phase B: DMA of k-step kt + 2 into cur ...
                dma(cur)
            if kt != nk - 1:
                read(oth, 0, (ti, kt + 1))                # ... and the next k-step's half 0
            par ^= 1
        if ti + 1 < my_tiles:
            read(par, 0, (ti + 1, 0))                     # behind the epilogue: first fragments of the next tile
    assert not unread[0] and not unread[1]


for mt in (1, 2, 3, 5):
    for nk_ in (8, 9, 16, 64):
        run_schedule(mt, nk_)
print("schedule: every read finds its (tile, k-step), no DMA overwrites unread data, nothing is left unread")
