#!/usr/bin/env python3
"""Generates blur16_asm.inc: the hand-scheduled row loop of blur16_stream (sift.hip) as gfx950 assembly, one inline-asm block per
(R, BGR, DS) variant.  The HIP kernel around it computes the wave's strip / segment, its LDS and row offsets and hands them over in
registers; everything from the first source row to the last stored row runs in the block below.

    python gen_blur16_asm.py > blur16_asm.inc          (build.py regenerates the file when this script is newer)

Arithmetic (oracle/oracle_sift.c gauss_blur16 = OpenCV's RowFilter<short,float> + SymmColumnFilter<Cast<float,short>>), every
product and sum rounded separately:
    row     t(x) = k[0] S(x-R);  t(x) += k[i] S(x-R+i), i = 1..2R          column  s = k[R] t(y);  s += k[R+j] (t(y+j) + t(y-j)), j = 1..R
    out = round-half-even(s) as 16 bits.

One wave owns a strip of 256 columns (4 per lane) and walks down its segment; per step one source row arrives and one output row
leaves.  What the hand schedule does that the compiler's did not:
  * the row pass of an output pair (p, p+1) runs as acc += (k[t], k[t-1]) * (e, e), t = 0..2R+1 with k[-1] = k[2R+1] = 0: both
    halves of the packed multiply read the SAME window register (op_sel broadcast), so no register pair is ever misaligned and no
    v_mov is needed; the zero taps contribute +0 and x + 0 = x exactly, so every rounded product and sum is the reference's.
    Pairs (k[t], k[t-1]) for t > R are the swapped pairs of 2R+1-t (symmetric kernel): R+1 scalar pairs hold all taps;
  * vmcnt waits are COUNTED (gfx950 counts loads and stores in one counter: hipcc's vmcnt(0) waited every step for the store it
    had just issued, 39 % of the wave cycles were spent parked); the two stores-per-step patterns (with / without the decimated
    row) are compile-time per ring slot, the warm-up steps (no store yet) take a conservative wait;
  * the column pass of the PREVIOUS output row sits between the issue of the window's LDS reads and the row pass that consumes
    them (the ring holds 2R+2 rows), so the LDS round trip is covered by arithmetic of the same wave; the window itself rolls
    through 16 registers (reads of the next quad are issued as soon as a quad is consumed);
  * two independent accumulation chains alternate everywhere, products are formed one tap ahead of their sums;
  * rounding is s + 1.5 * 2^23 (integer in the low mantissa bits; s is in [0, 32767]: positive normalised taps, samples in
    [0, 255 * 48], so saturate_cast never saturates) and one v_perm_b32 packs two results;
  * row pointers are scalar and incremental (reflect-101 walk), partial strips mask their stores with a saved exec mask, the
    reflected columns right of the image come from extra halo lanes (no LDS patch round trip).
"""
import sys

NQB = 4                     # window quads resident (rolling)


class Gen:
    def __init__(self, R, bgr, ds):
        self.R, self.bgr, self.ds = R, bgr, ds
        self.NP = 2 * R + 2
        self.RA = (R + 3) // 4 * 4
        self.S = self.RA - R
        self.BW = 256 + 2 * self.RA
        self.WN = self.S + 2 * R + 4
        self.NQT = (self.WN + 3) // 4
        self.BUF = (self.BW + 8 + 63) & ~63
        self.lines = []
        # ---- fixed registers -------------------------------------------------------------------------------------------
        v = 16
        self.RAWN = 6 if bgr else 4
        self.RAW = [v, v + self.RAWN]; v += 2 * self.RAWN          # per slot: m0 m1 (m2) | h (h1)
        self.E = v; v += 4 * NQB                                     # window quads
        self.T = v; v += 12                                          # temporaries (6 pairs)
        self.RING = v; v += 4 * self.NP
        self.vmax = v - 1
        s = 36
        self.s_rp = s; s += 2
        self.s_dp = s; s += 2
        self.s_dsp = s; s += 2
        self.s_smask = s; s += 2
        self.s_mg = s; s += 2          # (1.5 * 2^23, perm selector)
        self.s_t64 = s; s += 2
        self.s_rowb = s; s += 1
        self.s_gy = s; s += 1
        self.s_dir = s; s += 1
        self.s_hm1 = s; s += 1
        self.s_dstr = s; s += 1
        self.s_dsstr = s; s += 1
        self.s_n = s; s += 1
        self.s_i = s; s += 1
        self.s_c = list(range(s, s + 4)); s += 4       # BGR constants 1868 9617 4899 (as floats) + spare
        assert s <= 60
        self.s_tap = 60                 # 32 dwords: pairs (k[t], k[t-1]), t = 0..R
        self.smax = 91
        self.uid = 0

    # ---- helpers -----------------------------------------------------------------------------------------------------------
    def o(self, s):
        self.lines.append(s)

    def ring(self, slot, c):
        r = self.RING + 4 * (slot % self.NP) + 2 * c
        return "v[%d:%d]" % (r, r + 1)

    def vp(self, r):
        return "v[%d:%d]" % (r, r + 1)

    def tap_row(self, t):
        """(sgpr pair, op_sel bit for the low result, for the high result) of the pair (k[t], k[t-1])"""
        R = self.R
        if t <= R:
            return "s[%d:%d]" % (self.s_tap + 2 * t, self.s_tap + 2 * t + 1), 0, 1
        u = 2 * R + 1 - t
        return "s[%d:%d]" % (self.s_tap + 2 * u, self.s_tap + 2 * u + 1), 1, 0

    def tap_bcast(self, j):
        """k[R - j] = k[R + j] broadcast: low half of pair R - j"""
        u = self.R - j
        return "s[%d:%d]" % (self.s_tap + 2 * u, self.s_tap + 2 * u + 1)

    def e_reg(self, q):
        """register pair holding window element q and which half"""
        g = q // 4
        base = self.E + 4 * (g % NQB) + (q % 4 // 2) * 2
        return "v[%d:%d]" % (base, base + 1), q & 1

    # ---- blocks of one step ------------------------------------------------------------------------------------------------
    def lds_read_quad(self, g, par):
        off = (par * self.BUF + 4 * g) * 4
        base = self.E + 4 * (g % NQB)
        self.o("ds_read_b128 v[%d:%d], %%0 offset:%d" % (base, base + 3, off))
        self.ldsq.append(("q", g))

    def wait_quad(self, g):
        idx = max(i for i, x in enumerate(self.ldsq) if x == ("q", g))
        if idx <= self.lds_done:
            return
        n = len(self.ldsq) - 1 - idx
        self.o("s_waitcnt lgkmcnt(%d)" % n)
        self.lds_done = idx

    def block_col(self, slot, lab):
        """column pass of the output row whose 2R+1 rows are the ring slots slot+1 .. slot-1, then the store(s)"""
        R, NP = self.R, self.NP
        cc = (slot + 1 + R) % NP
        s = [self.T + 0, self.T + 2]                  # sums of chains 0, 1
        tt = [[self.T + 4, self.T + 6], [self.T + 8, self.T + 10]]   # tt[chain][jj & 1]
        for c in (0, 1):
            self.o("v_pk_mul_f32 %s, %s, %s op_sel_hi:[0,1]" % (self.vp(s[c]), self.tap_bcast(0), self.ring(cc, c)))
        for g in range(1, R + 3):
            if 1 <= g - 2 <= R:
                for c in (0, 1):
                    self.o("v_pk_add_f32 %s, %s, %s" % (self.vp(s[c]), self.vp(s[c]), self.vp(tt[c][(g - 2) & 1])))
            if g <= R:
                for c in (0, 1):
                    self.o("v_pk_add_f32 %s, %s, %s" % (self.vp(tt[c][g & 1]), self.ring(cc + g, c), self.ring(cc - g, c)))
            if 1 <= g - 1 <= R:
                for c in (0, 1):
                    t = self.vp(tt[c][(g - 1) & 1])
                    self.o("v_pk_mul_f32 %s, %s, %s op_sel_hi:[0,1]" % (t, self.tap_bcast(g - 1), t))
        mg = "s[%d:%d]" % (self.s_mg, self.s_mg + 1)
        for c in (0, 1):
            self.o("v_pk_add_f32 %s, %s, %s op_sel_hi:[1,0]" % (self.vp(s[c]), self.vp(s[c]), mg))
        o0 = self.T + 4
        self.o("v_perm_b32 v%d, v%d, v%d, s%d" % (o0, s[0] + 1, s[0], self.s_mg + 1))
        self.o("v_perm_b32 v%d, v%d, v%d, s%d" % (o0 + 1, s[1] + 1, s[1], self.s_mg + 1))
        self.o("s_mov_b64 exec, s[%d:%d]" % (self.s_smask, self.s_smask + 1))
        self.o("global_store_dwordx2 %%5, v[%d:%d], s[%d:%d]" % (o0, o0 + 1, self.s_dp, self.s_dp + 1))
        nst = 1
        if self.ds and (slot % 2 == 1):               # output row r = i - 1 - 2R is even when the step (and its slot) is odd
            self.o("v_perm_b32 v%d, v%d, v%d, s%d" % (o0 + 2, o0 + 1, o0, self.s_mg + 1))
            self.o("global_store_dword %%6, v%d, s[%d:%d]" % (o0 + 2, self.s_dsp, self.s_dsp + 1))
            nst = 2
        self.o("s_mov_b64 exec, -1")
        self.o("s_add_u32 s%d, s%d, s%d" % (self.s_dp, self.s_dp, self.s_dstr))
        self.o("s_addc_u32 s%d, s%d, 0" % (self.s_dp + 1, self.s_dp + 1))
        if nst == 2:
            self.o("s_add_u32 s%d, s%d, s%d" % (self.s_dsp, self.s_dsp, self.s_dsstr))
            self.o("s_addc_u32 s%d, s%d, 0" % (self.s_dsp + 1, self.s_dsp + 1))
        return nst

    def gray48(self, dst, src_bytes):
        """dst = float(((1868 b + 9617 g + 4899 r + 8192) >> 14) * 48), exact in f32 (all integers < 2^24); src_bytes = 3 x (vreg, byte)"""
        tb, tg, tr = self.T + 6, self.T + 7, self.T + 8
        for t, (r, b) in zip((tb, tg, tr), src_bytes):
            self.o("v_cvt_f32_ubyte%d v%d, v%d" % (b, t, r))
        self.o("v_mov_b32 v%d, 0x46000000" % dst)                       # 8192.0
        self.o("v_fmac_f32 v%d, s%d, v%d" % (dst, self.s_c[0], tb))
        self.o("v_fmac_f32 v%d, s%d, v%d" % (dst, self.s_c[1], tg))
        self.o("v_fmac_f32 v%d, s%d, v%d" % (dst, self.s_c[2], tr))
        self.o("v_mul_f32 v%d, 0x38800000, v%d" % (dst, dst))          # 2^-14
        self.o("v_floor_f32 v%d, v%d" % (dst, dst))
        self.o("v_mul_f32 v%d, 0x42400000, v%d" % (dst, dst))          # 48.0

    def load_row(self, rslot):
        """issue the loads of the next source row into raw slot rslot, advance the reflect-101 walk"""
        raw = self.RAW[rslot]
        rp = "s[%d:%d]" % (self.s_rp, self.s_rp + 1)
        if self.bgr:
            self.o("global_load_dwordx3 v[%d:%d], %%3, %s" % (raw, raw + 2, rp))
            self.o("global_load_dwordx2 v[%d:%d], %%4, %s" % (raw + 4, raw + 5, rp))
        else:
            self.o("global_load_dwordx2 v[%d:%d], %%3, %s" % (raw, raw + 1, rp))
            self.o("global_load_ushort v%d, %%4, %s" % (raw + 2, rp))
        gy, d, t = self.s_gy, self.s_dir, self.s_t64
        self.o("s_cmp_eq_u32 s%d, 0" % gy)
        self.o("s_cselect_b32 s%d, 1, s%d" % (d, d))
        self.o("s_cmp_ge_i32 s%d, s%d" % (gy, self.s_hm1))
        self.o("s_cselect_b32 s%d, -1, s%d" % (d, d))
        self.o("s_add_i32 s%d, s%d, s%d" % (gy, gy, d))
        self.o("s_mul_i32 s%d, s%d, s%d" % (t, d, self.s_rowb))
        self.o("s_ashr_i32 s%d, s%d, 31" % (t + 1, t))
        self.o("s_add_u32 s%d, s%d, s%d" % (self.s_rp, self.s_rp, t))
        self.o("s_addc_u32 s%d, s%d, s%d" % (self.s_rp + 1, self.s_rp + 1, t + 1))

    def put_row(self, rslot, par):
        """raw slot -> floats -> LDS row buffer `par` (main quad of the lane + one halo sample)"""
        raw = self.RAW[rslot]
        m = self.T                                   # 4 floats + halo in T+0..T+4
        if self.bgr:
            by = [(raw, 0), (raw, 1), (raw, 2), (raw, 3), (raw + 1, 0), (raw + 1, 1), (raw + 1, 2), (raw + 1, 3),
                  (raw + 2, 0), (raw + 2, 1), (raw + 2, 2), (raw + 2, 3)]
            for p in range(4):
                self.gray48(m + p, by[3 * p:3 * p + 3])
            # halo: 8 bytes from the aligned address, shifted down by %7 bits -> b g r in the low three bytes
            self.o("v_lshrrev_b64 v[%d:%d], %%7, v[%d:%d]" % (raw + 4, raw + 5, raw + 4, raw + 5))
            self.gray48(m + 4, [(raw + 4, 0), (raw + 4, 1), (raw + 4, 2)])
        else:
            self.o("v_cvt_f32_i32_sdwa v%d, sext(v%d) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0" % (m, raw))
            self.o("v_cvt_f32_i32_sdwa v%d, sext(v%d) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1" % (m + 1, raw))
            self.o("v_cvt_f32_i32_sdwa v%d, sext(v%d) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0" % (m + 2, raw + 1))
            self.o("v_cvt_f32_i32_sdwa v%d, sext(v%d) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1" % (m + 3, raw + 1))
            self.o("v_cvt_f32_i32_sdwa v%d, sext(v%d) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0" % (m + 4, raw + 2))
        off = par * self.BUF * 4
        self.o("ds_write_b128 %%1, v[%d:%d] offset:%d" % (m, m + 3, off))
        self.o("ds_write_b32 %%2, v%d offset:%d" % (m + 4, off))
        self.ldsq.append(("w", 0)); self.ldsq.append(("w", 1))

    def block_row(self, slot, par):
        """row pass of the arriving row: window quads consumed in ascending order, result -> ring[slot]"""
        R, S, WN = self.R, self.S, self.WN
        acc = [self.ring(slot, 0), self.ring(slot, 1)]               # accumulate straight into the ring slot
        P = [[self.T + 4, self.T + 6], [self.T + 8, self.T + 10]]    # P[chain][q & 1]
        pending = []                                                 # adds of the previous q

        def taps_at(q):
            out = []
            for c in (0, 1):
                t = q - S - 2 * c
                if 0 <= t <= 2 * R + 1:
                    out.append((c, t))
            return out

        last_q_of_quad = {}
        for q in range(S, WN):
            if taps_at(q):
                last_q_of_quad[q // 4] = q
        for q in range(S, WN):
            use = taps_at(q)
            if not use:
                continue
            self.wait_quad(q // 4)
            er, half = self.e_reg(q)
            adds = []
            for c, t in use:
                sp, lo, hi = self.tap_row(t)
                dst = acc[c] if t == 0 else self.vp(P[c][q & 1])
                self.o("v_pk_mul_f32 %s, %s, %s op_sel:[%d,%d] op_sel_hi:[%d,%d]" % (dst, sp, er, lo, half, hi, half))
                if t != 0:
                    adds.append("v_pk_add_f32 %s, %s, %s" % (acc[c], acc[c], self.vp(P[c][q & 1])))
            for a in pending:
                self.o(a)
            pending = adds
            g = q // 4
            if last_q_of_quad.get(g) == q and g + NQB < self.NQT:
                self.lds_read_quad(g + NQB, par)
        for a in pending:
            self.o(a)

    def step(self, slot, lab):
        """one row step.  slot = ring slot of the arriving row = step index mod NP; LDS buffer parity = slot & 1"""
        R = self.R
        par = slot & 1
        self.ldsq, self.lds_done = [], -1
        self.o("// ---- step, ring slot %d" % slot)
        self.o("s_cmp_ge_u32 s%d, s%d" % (self.s_i, self.s_n))
        self.o("s_cbranch_scc1 L_end_%=")
        for g in range(min(NQB, self.NQT)):
            self.lds_read_quad(g, par)
        # column pass of the previous output row (skipped while the ring fills)
        self.o("s_cmp_lt_u32 s%d, %d" % (self.s_i, 2 * R + 1))
        self.o("s_cbranch_scc1 L_nocol_%d_%%=" % lab)
        self.block_col(slot, lab)
        self.o("L_nocol_%d_%%=:" % lab)
        # next row: raw slot (slot + 1) & 1 -> LDS buffer par ^ 1, then reload the slot with the row after next
        # vmcnt: in order, a step issues [store (, decimated store)] [2 loads].  The slot was loaded two steps ago; after its loads
        # came: stores of the previous step, its 2 loads, the stores of this step = 4 (5 with the decimated row: exactly one of two
        # consecutive steps has it).  While the ring fills there are no stores: only the previous step's 2 loads are younger.
        n_steady = 5 if self.ds else 4
        self.o("s_cmp_lt_u32 s%d, %d" % (self.s_i, 2 * R + 2))
        self.o("s_cbranch_scc1 L_wc_%d_%%=" % lab)
        self.o("s_waitcnt vmcnt(%d)" % n_steady)
        self.o("s_branch L_wd_%d_%%=" % lab)
        self.o("L_wc_%d_%%=:" % lab)
        self.o("s_waitcnt vmcnt(2)")
        self.o("L_wd_%d_%%=:" % lab)
        rslot = (slot + 1) & 1
        self.put_row(rslot, par ^ 1)
        self.load_row(rslot)
        self.block_row(slot, par)
        self.o("s_add_u32 s%d, s%d, 1" % (self.s_i, self.s_i))

    def body(self):
        R = self.R
        o = self.o
        o("// inputs: %0 lds window  %1 lds main  %2 lds halo  %3 main offset  %4 halo offset  %5 dst offset  %6 decimated offset" + ("  %7 halo shift" if self.bgr else ""))
        B = 8 if self.bgr else 7
        names = ["rp", "dp", "dsp", "kp", "smask", "rowb", "gy", "dir", "hm1", "dstr", "n"]
        idx = {n: B + i for i, n in enumerate(names)}
        o("s_mov_b64 s[%d:%d], %%%d" % (self.s_rp, self.s_rp + 1, idx["rp"]))
        o("s_mov_b64 s[%d:%d], %%%d" % (self.s_dp, self.s_dp + 1, idx["dp"]))
        o("s_mov_b64 s[%d:%d], %%%d" % (self.s_dsp, self.s_dsp + 1, idx["dsp"]))
        o("s_mov_b64 s[%d:%d], %%%d" % (self.s_t64, self.s_t64 + 1, idx["kp"]))
        o("s_mov_b64 s[%d:%d], %%%d" % (self.s_smask, self.s_smask + 1, idx["smask"]))
        o("s_mov_b32 s%d, %%%d" % (self.s_rowb, idx["rowb"]))
        o("s_mov_b32 s%d, %%%d" % (self.s_gy, idx["gy"]))
        o("s_mov_b32 s%d, %%%d" % (self.s_dir, idx["dir"]))
        o("s_mov_b32 s%d, %%%d" % (self.s_hm1, idx["hm1"]))
        o("s_mov_b32 s%d, %%%d" % (self.s_dstr, idx["dstr"]))
        o("s_lshr_b32 s%d, s%d, 1" % (self.s_dsstr, self.s_dstr))
        o("s_mov_b32 s%d, %%%d" % (self.s_n, idx["n"]))
        o("s_mov_b32 s%d, 0" % self.s_i)
        o("s_mov_b32 s%d, 0x4b400000" % self.s_mg)                     # 1.5 * 2^23
        o("s_mov_b32 s%d, 0x05040100" % (self.s_mg + 1))               # v_perm_b32: low halves of two dwords
        if self.bgr:
            o("s_mov_b32 s%d, 0x44e98000" % self.s_c[0])               # 1868.0
            o("s_mov_b32 s%d, 0x46164400" % self.s_c[1])               # 9617.0
            o("s_mov_b32 s%d, 0x45991800" % self.s_c[2])               # 4899.0
        o("s_load_dwordx16 s[%d:%d], s[%d:%d], 0" % (self.s_tap, self.s_tap + 15, self.s_t64, self.s_t64 + 1))
        o("s_load_dwordx16 s[%d:%d], s[%d:%d], 64" % (self.s_tap + 16, self.s_tap + 31, self.s_t64, self.s_t64 + 1))
        self.ldsq, self.lds_done = [], -1
        self.load_row(0)
        self.load_row(1)
        o("s_waitcnt vmcnt(2) lgkmcnt(0)")
        self.put_row(0, 0)
        self.load_row(0)
        o("L_loop_%=:")
        for slot in range(self.NP):
            self.step(slot, slot)
        o("s_branch L_loop_%=")
        o("L_end_%=:")
        o("s_waitcnt vmcnt(0) lgkmcnt(0)")

    def emit(self, out):
        self.body()
        name = "blur16_asm_r%d%s%s" % (self.R, "_bgr" if self.bgr else "", "_ds" if self.ds else "")
        out.write("// R = %d%s%s: ring %d slots, %d window quads, VGPRs v16..v%d, SGPRs s36..s%d\n" % (self.R, ", BGR source" if self.bgr else "", ", decimated copy" if self.ds else "", self.NP, self.NQT, self.vmax, self.smax))
        out.write("__device__ __forceinline__ void %s(unsigned lds_win, unsigned lds_main, unsigned lds_halo, unsigned moff, unsigned hoff, unsigned doff, unsigned dsoff,%s\n" % (name, " unsigned hsh," if self.bgr else ""))
        out.write("        unsigned long long rp, unsigned long long dp, unsigned long long dsp, unsigned long long kp, unsigned long long smask, int rowb, int gy, int dir, int hm1, int dstr, int n) {\n")
        out.write("    asm volatile(\n")
        for l in self.lines:
            if l.startswith("//"):
                out.write("        %s\n" % l)
            else:
                out.write('        "%s\\n"\n' % l)
        vin = '"v"(lds_win), "v"(lds_main), "v"(lds_halo), "v"(moff), "v"(hoff), "v"(doff), "v"(dsoff)' + (', "v"(hsh)' if self.bgr else "")
        sin = '"s"(rp), "s"(dp), "s"(dsp), "s"(kp), "s"(smask), "s"(rowb), "s"(gy), "s"(dir), "s"(hm1), "s"(dstr), "s"(n)'
        out.write("        :\n        : %s,\n          %s\n" % (vin, sin))
        clob = ['"v%d"' % r for r in range(16, self.vmax + 1)] + ['"s%d"' % r for r in range(36, self.smax + 1)] + ['"vcc"', '"scc"', '"memory"']
        out.write("        : ")
        for i in range(0, len(clob), 24):
            out.write(("          " if i else "") + ", ".join(clob[i:i + 24]) + (",\n" if i + 24 < len(clob) else ");\n"))
        out.write("}\n")
        out.write("constexpr int %s_vgprs = %d;\n\n" % (name, self.vmax + 1))


VARIANTS = [(5, False, False), (6, False, False), (8, False, False), (8, False, True), (10, False, False), (13, False, False), (6, True, False)]


def main():
    out = sys.stdout
    out.write("// GENERATED by gen_blur16_asm.py -- do not edit.  The row loop of blur16_stream as hand-scheduled gfx950 assembly.\n\n")
    for R, bgr, ds in VARIANTS:
        Gen(R, bgr, ds).emit(out)


if __name__ == "__main__":
    main()
