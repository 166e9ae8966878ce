#!/usr/bin/env python3
"""Generator of swin_mlp512_loop.inc: the body of the C = 512 fused Swin MLP kernel (swin_mlp512.hip) as ONE inline-asm statement
with hand-assigned registers.

Why asm: the kernel needs the whole register file of a wave (oacc 256 AGPRs, the rows' xb 128 VGPRs, ~100 more) and the
compiler's allocator does not survive that.  As HIP source (builtin MFMAs, then asm MFMAs with tied accumulators) it came out
with 400-560 spilled registers: MFMA accumulators are untied from their C operand, the kernel-wide MFMA form puts GEMM 1's
accumulators into AGPRs that do not exist any more, the xb operands lived in scratch; with the loop alone in asm and oacc
handed back as 64 "=a" operands the loop was clean (0 spills) and the LayerNorm epilogue behind it did not finish compiling
in ten minutes.  So the compiler keeps what it is good at -- operand set-up and the bias rows' copy into LDS -- and this
script assigns every register of everything else:
    a[0:255]    oacc[mt][jo] = a[4 (32 mt + jo) ..]          (literal, clobbered)
    v[0:127]    xf[mt][ks] = v[4 (16 mt + ks) ..] in the loop, the residual rows' quads in the epilogue   (literal, clobbered)
    v[144:255]  temporaries (map below)                       (literal, clobbered)
    operands    "+v" w1p[4], w2p (fragment read bases, toggled between the ring slots), b1p (bias pointer);
                "v"  v1[4], v2 (LDS-DMA lane offsets), vecp (LDS address of b2 | gamma | beta + 32 quad), xboff (row * 1024 + 16 quad),
                     bp16, bp32 (ds_bpermute addresses of lane ^ 16, lane ^ 32);
                "s"  w1r, w2r, xr, xbr (buffer descriptors: rows past m lie past the extent -- loads give zeros, stores are dropped),
                     swave = wave * 1024 (source side), sldsw = LDS base + wave * 1024 (destination side), eps
                     dbgr / dbgoff / bid (timing build), attr / wpr (PROJ: the attention output rows, attn.proj.weight)
    s[40:99] scratch scalars, m0 saved / restored, scc; vcc and exec untouched
The variants (VARIANTS below; template parameter V of swin_mlp512_kernel): 0 the MLP block; 1 PROJ: attention projection + LayerNorm +
residual in front of it (proj_gemm, proj_ln); 9 QKV: PROJ + the NEXT block's qkv Linear behind it (qkv_phase) -- no operand was left for
it (16 VGPRs and ~40 SGPRs survive the clobber list), so the kernel passes Wqkv's descriptor as `dbgr` and the qkv rows' as `xbr`
(the variant writes no shadow and no timing data) and the row offset is derived from xboff; 2 .. 8 ablations and the cycle-counter
build of the loop, 10 .. (VSC_GEN_QKV_ABL=1) those of the QKV phase: diagnostic builds only (-DVSC_MLP_ABLATION).
Wait states the compiler would pad and an asm statement must carry itself (cdna_hip_programming.md 5.7):
    MFMA D (VGPR) -> vector reader: GEMM 1's last MFMA is followed by s_nop 7 + the loop head (>= 12 states);
    MFMA D (AGPR) -> v_accvgpr_read: s_nop 15 behind the last MFMA;
    vector write -> MFMA operand: s_nop 1 between the GELU's last v_cvt_pk (hf) and GEMM 2's first MFMA;
    v_accvgpr_write -> MFMA C: the bias initialisation is hundreds of instructions ahead of the first GEMM 2 MFMA;
    MFMA chain (D taken whole as the next C): none;  16-byte store -> overwrite of its data: register sets alternate per column group.
LDS reads return in order: the script tracks the queue and emits counted s_waitcnt lgkmcnt(n).

    python3 gen_mlp512_loop.py > swin_mlp512_loop.inc
"""
import struct
import sys

SLOT = 32768
LDS_W1, LDS_W2 = 0, 2 * SLOT
NCH = 64
PF, STAGGER, TIMING, ABL, PROJ, QKV = 8, 0, False, 0, False, False    # defaults; main() builds the variants listed in VARIANTS
# ABL (diagnostic variants, wrong results): 1 no GELU arithmetic, 2 no LDS-DMA in the loop, 4 no fragment reads, 8 no MFMAs in the loop;
# in the QKV phase: 16 no LDS-DMA, 32 no barrier, 64 no bias + rounding + store, 128 no fragment reads, 1024 cycle stamps, 4096 no stores,
# 8192 a fragment ring of 10, 16384 ring refills on every other group
GELU_DEG = 8
GELU_U = 4.5
GELU_ZS = 2.0 / (4.5 * 4.5)
GELU_C = [1.569020897e-01, -7.717858255e-02, 5.482625961e-02, -4.047540203e-02, 2.754251473e-02, -1.697185636e-02, 1.244884357e-02,
          -9.152771905e-03, 3.170517040e-03]   # gelu_poly.h
NSTEP = GELU_DEG + 3


def f32(x):
    return "0x%08x" % struct.unpack("<I", struct.pack("<f", x))[0]


# ---- register map -------------------------------------------------------------------------------------------------------------
T = 144
PFMAX = 10
def vq(b): return f"v[{b}:{b + 3}]"
def vp(b): return f"v[{b}:{b + 1}]"
def HACC(mt, j): return T + 4 * (2 * mt + j)          # 144..159 (the GELU works in place here)
def WQ(k): return T + 16 + 4 * k                        # 160..199: up to PFMAX fragment quads
def HF(mt): return T + 56 + 4 * mt                      # 200..207
def HFN(mt): return T + 64 + 4 * mt                     # 208..215
def BZ(j): return T + 72 + 4 * j                        # 216..223
GT, GZ, GQ = T + 80, T + 88, T + 96                     # 224.., 232.., 240..  (four pairs each)
VU, VC7 = T + 104, T + 106                              # 248 (+U), 250:251 (c7, low half used)
S_I, S_T, S_T2, S_SLOT, S_M0, S_SO1, S_SO2, S_NU = 40, 41, 42, 43, 44, 45, 46, 47
S_ZS, S_C8 = 48, 50
def S_C(k): return 52 + 2 * k                           # k = 0..6 -> s[52:65]
def sp(b): return f"s[{b}:{b + 1}]"

def ACC(mt, jo): return f"a[{4 * (32 * mt + jo)}:{4 * (32 * mt + jo) + 3}]"
def ACCR(mt, jo, r): return f"a{4 * (32 * mt + jo) + r}"
def XF(mt, ks): return vq(4 * (16 * mt + ks))
def W1P(a): return f"%[w1p{a}]"
W2P, B1P, V2, W1R, W2R, SW, SLW = "%[w2p]", "%[b1p]", "%[v2]", "%[w1r]", "%[w2r]", "%[swave]", "%[sldsw]"
VECP, XBOFF, BP16, BP32, XR, XBR, EPS = "%[vecp]", "%[xboff]", "%[bp16]", "%[bp32]", "%[xr]", "%[xbr]", "%[eps]"
ATTR, WPR = "%[attr]", "%[wpr]"      # PROJ: descriptors of the attention output rows and of attn.proj.weight
WQR, QR = "%[dbgr]", "%[xbr]"   # QKV (no operands left: 16 VGPRs outside the clobbers, SGPRs likewise): the kernel passes the next block's qkv weight in the timing buffer's place and the qkv rows in the shadow's; row * 3072 + 16 quad is derived from xboff
DBGR, DBGOFF = "%[dbgr]", "%[dbgoff]"      # timing variant: per-wave cycle counters go to dbgr at byte offset dbgoff
S_TS, S_ACC = 70, 80                     # s[70:79] time stamps (pairs), s80.. accumulated differences
S_MT1X, S_MT1B = 66, 67     # byte offsets of the second 16-row tile in x / xb
S_SLOT2 = 68                # LDS offset of the W2 slot being refilled (S_SLOT: the W1 slot)
S_LB1, S_LB2 = 69, 86       # m0 bases of the two refills (slot base + LDS base + this wave's 1 KiB)
BID = "%[bid]"
def V1(a): return f"%[v1{a}]"


class Emit:
    def __init__(self):
        self.lines = []
        self.q = []          # outstanding LDS reads (tags), oldest first
        self.vq = []         # outstanding vector-memory operations tracked for counted waits (proj_ln)

    def i(self, s):
        self.lines.append(s)

    def c(self, s):
        self.lines.append("; " + s)

    def ds_read(self, tag, dst, addr, off):
        self.i(f"ds_read_b128 {vq(dst)}, {addr}" + (f" offset:{off}" if off else ""))
        self.q.append(tag)
        assert len(self.q) <= 15

    def wait(self, tag):
        if tag not in self.q:
            return
        k = self.q.index(tag)
        n = len(self.q) - 1 - k
        self.i(f"s_waitcnt lgkmcnt({n})")
        self.q = self.q[k + 1:]

    def wait_all(self):
        if self.q:
            self.i("s_waitcnt lgkmcnt(0)")
        self.q = []

    # vector-memory operations (loads and stores retire in issue order on this part: the compiler's own counted vmcnt waits rely on it)
    def vm(self, tag, text):
        self.i(text)
        self.vq.append(tag)
        assert len(self.vq) <= 63

    def vm_wait(self, tag):
        if tag not in self.vq:
            return
        k = self.vq.index(tag)
        self.i(f"s_waitcnt vmcnt({len(self.vq) - 1 - k})")
        self.vq = self.vq[k + 1:]

    def vm_wait_all(self):
        self.i("s_waitcnt vmcnt(0)")
        self.vq = []


def frag_addr(g):
    """(base register, immediate) of the fragment of MFMA group g: 0..31 GEMM 2 (output tile g), 32..63 GEMM 1 (j = e & 1, ks = e >> 1)"""
    if g < 32:
        return W2P, 2048 * (g >> 1) + 256 * (g & 1)
    e = g - 32
    j, ks = e & 1, e >> 1
    return W1P(ks & 3), 16384 * j + 256 * (ks >> 2)


def group(E, g, end, between=(), after=(), zero_start=False):
    """the two MFMAs of group g, then the request of the fragment PF groups ahead (none at or beyond `end`).  `between`: up to three
    scalar / memory instructions issued in the shadow of the first MFMA (an MFMA occupies the matrix pipe for 16 cycles and one issue
    slot of 4: three more instructions of other kinds fit behind it for free -- put behind the SECOND MFMA they delay the next
    group's first one); `after`: instructions behind the fragment request."""
    E.wait(("f", g))
    w = vq(WQ(g % PF))
    first, second = [], []
    if ABL & 8:
        pass
    elif g < 32:
        first.append(f"v_mfma_f32_16x16x32_bf16 {ACC(0, g)}, {w}, {vq(HF(0))}, {ACC(0, g)}")
        second.append(f"v_mfma_f32_16x16x32_bf16 {ACC(1, g)}, {w}, {vq(HF(1))}, {ACC(1, g)}")
    else:
        e = g - 32
        j, ks = e & 1, e >> 1
        for mt, lst in ((0, first), (1, second)):
            c = ("0" if zero_start else vq(BZ(j))) if ks == 0 else vq(HACC(mt, j))     # a chain starts from the bias of its four hidden units (or from zero)
            lst.append(f"v_mfma_f32_16x16x32_bf16 {vq(HACC(mt, j))}, {w}, {XF(mt, ks)}, {c}")
    for l in first:
        E.i(l)
    for l in between:
        E.i(l)
    for l in second:
        E.i(l)
    if g + PF < end and not (ABL & 4) and not (ABL & 128 and zero_start):
        a, off = frag_addr(g + PF)
        E.ds_read(("f", g + PF), WQ(g % PF), a, off)
    for l in after:
        E.i(l)


def dma_items(which):
    """LDS-DMA of one chunk as 8 items ([scalar set-up], [the load]).  which = 'w1': chunk i + 2 (soffset base S_SO1) -> W1 slot i & 1
    (past the last chunk the source lies past the descriptor's extent: zeros arrive, nobody reads them); 'w2': chunk i + 1 (S_SO2) -> W2
    slot (i + 1) & 1.  m0 = LDS destination of the instruction (wave-uniform: S_LB1 / S_LB2 = slot base + this wave's 1 KiB), lane l
    lands at m0 + 16 l."""
    so = S_SO1 if which == "w1" else S_SO2
    lb = S_LB1 if which == "w1" else S_LB2
    rs = W1R if which == "w1" else W2R
    items = []
    for qq in range(8):
        voff = V1(qq & 3) if which == "w1" else V2
        items.append(([f"s_add_u32 m0, s{lb}, {qq * 4096}", f"s_add_u32 s{S_T2}, s{so}, {qq * 4096}"],
                      [f"buffer_load_dwordx4 {voff}, {rs}, s{S_T2} offen lds"]))
    return items


def dma_bases(E):
    """LDS destination bases of this iteration's refills: W1 slot S_SLOT, W2 slot S_SLOT2, + LDS base + wave * 1024"""
    E.i(f"s_add_u32 s{S_LB1}, s{S_SLOT}, {SLW}")
    E.i(f"s_add_u32 s{S_LB2}, s{S_SLOT2}, {SLW}")
    E.i(f"s_add_u32 s{S_LB2}, s{S_LB2}, {LDS_W2}")


def dma(E, which):
    for pre, load in dma_items(which):
        for l in pre + load:
            E.i(l)


def gelu(E, mt, fill):
    """GELU + bf16 rounding of the 16-row tile mt (the fc1 bias is already in: GEMM 1's chains start from it): 8 values per lane, in place in
    HACC(mt, 0..1) -> HF(mt); fill(step) after each step"""
    x = [HACC(mt, 0), HACC(mt, 0) + 2, HACC(mt, 1), HACC(mt, 1) + 2]    # four pairs
    t = [GT + 2 * k for k in range(4)]
    z = [GZ + 2 * k for k in range(4)]
    q = [GQ + 2 * k for k in range(4)]
    if ABL & 1:
        for k in range(4):
            E.i(f"v_cvt_pk_bf16_f32 v{HF(mt) + k}, v{x[k]}, v{x[k] + 1}")
        return
    for k in range(4):
        for h in range(2):
            E.i(f"v_med3_f32 v{t[k] + h}, v{x[k] + h}, s{S_NU}, v{VU}")
    fill(0)
    for k in range(4):
        E.i(f"v_pk_mul_f32 {vp(z[k])}, {vp(t[k])}, {vp(t[k])}")
    for k in range(4):
        E.i(f"v_pk_fma_f32 {vp(z[k])}, {vp(z[k])}, {sp(S_ZS)}, -1.0 op_sel_hi:[1,0,0]")
    fill(1)
    for k in range(4):
        E.i(f"v_pk_fma_f32 {vp(q[k])}, {vp(z[k])}, {sp(S_C8)}, {vp(VC7)} op_sel_hi:[1,0,0]")
    fill(2)
    for c in range(GELU_DEG - 2, -1, -1):
        for k in range(4):
            E.i(f"v_pk_fma_f32 {vp(q[k])}, {vp(q[k])}, {vp(z[k])}, {sp(S_C(c))} op_sel_hi:[1,1,0]")
        fill(GELU_DEG + 1 - c)
    for k in range(4):
        E.i(f"v_pk_fma_f32 {vp(q[k])}, {vp(t[k])}, {vp(q[k])}, 0.5 op_sel_hi:[1,1,0]")
    for k in range(4):
        E.i(f"v_pk_mul_f32 {vp(x[k])}, {vp(x[k])}, {vp(q[k])}")
    fill(NSTEP - 1)
    for k in range(4):
        E.i(f"v_cvt_pk_bf16_f32 v{HF(mt) + k}, v{x[k]}, v{x[k] + 1}")


def stamp(E, k):
    """timing variant: s[S_TS + 2k : +1] = s_memtime (an SMEM return: the LDS queue must be empty here, and is drained)"""
    if not TIMING:
        return
    assert not E.q
    E.i(f"s_memtime {sp(S_TS + 2 * k)}")
    E.i("s_waitcnt lgkmcnt(0)")


def lap(E, acc, k1, k0):
    """timing variant: s[S_ACC + acc] += stamp k1 - stamp k0 (low words)"""
    if not TIMING:
        return
    E.i(f"s_sub_u32 s{S_T}, s{S_TS + 2 * k1}, s{S_TS + 2 * k0}")
    E.i(f"s_add_u32 s{S_ACC + acc}, s{S_ACC + acc}, s{S_T}")


def SC1():
    """PROJ: the residual rows the epilogue reads were stored by this same wave earlier in the launch (x1): read them at agent scope
    (past the CU's vector cache, which holds the lines of x the projection phase loaded)"""
    return " sc1" if PROJ else ""


def col_off(jo):
    """byte offset (fp32 rows) of the lane's four columns of output tile jo, without the 32 quad part: columns 32 (jo >> 1) + 4 (jo & 1) + 8 quad + r"""
    return 128 * (jo >> 1) + 16 * (jo & 1)


def reduce4(E, v, tmp):
    """v += v of lane ^ 16, then of lane ^ 32: the four lanes (fr, 0..3) that hold one row"""
    for bp in (BP16, BP32):
        E.i(f"ds_bpermute_b32 v{tmp}, {bp}, v{v}")
        E.i("s_waitcnt lgkmcnt(0)")
        E.i(f"v_add_f32 v{v}, v{v}, v{tmp}")


def ln_stats(E, MEAN, RSTD, S, TMP, P0, P1, W):
    """row statistics of the wave's two 16-row tiles over the accumulator registers: two passes, a row sits in four lanes"""
    for mt in range(2):
        E.c(f"statistics of tile {mt}: two passes over the registers")
        for pas in range(2):
            E.i(f"v_mov_b32 v{P0}, 0")
            E.i(f"v_mov_b32 v{P0 + 1}, 0")
            E.i(f"v_mov_b32 v{P1}, 0")
            E.i(f"v_mov_b32 v{P1 + 1}, 0")
            for jo in range(32):
                w = W + 4 * (jo & 1)          # two work quads, alternating
                for r in range(4):
                    E.i(f"v_accvgpr_read_b32 v{w + r}, {ACCR(mt, jo, r)}")
                if pas == 0:
                    E.i(f"v_pk_add_f32 {vp(P0)}, {vp(P0)}, {vp(w)}")
                    E.i(f"v_pk_add_f32 {vp(P1)}, {vp(P1)}, {vp(w + 2)}")
                else:
                    E.i(f"v_pk_add_f32 {vp(w)}, {vp(w)}, {vp(MEAN[mt])} op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]")
                    E.i(f"v_pk_add_f32 {vp(w + 2)}, {vp(w + 2)}, {vp(MEAN[mt])} op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]")
                    E.i(f"v_pk_fma_f32 {vp(P0)}, {vp(w)}, {vp(w)}, {vp(P0)}")
                    E.i(f"v_pk_fma_f32 {vp(P1)}, {vp(w + 2)}, {vp(w + 2)}, {vp(P1)}")
            E.i(f"v_pk_add_f32 {vp(P0)}, {vp(P0)}, {vp(P1)}")
            E.i(f"v_add_f32 v{S}, v{P0}, v{P0 + 1}")
            reduce4(E, S, TMP)
            if pas == 0:
                E.i(f"v_mul_f32 v{MEAN[mt]}, {f32(1.0 / 512)}, v{S}")
            else:
                E.i(f"v_mov_b32 v{TMP}, {EPS}")
                E.i(f"v_fmac_f32 v{TMP}, {f32(1.0 / 512)}, v{S}")
                E.i(f"v_rsq_f32 v{RSTD[mt]}, v{TMP}")
                E.i("s_nop 1")


def proj_gemm(E):
    """PROJ: oacc = att Wp^T + bp for the wave's 32 rows, as sixteen 32-column chunks on GEMM 1's machinery (the attention output rows
    are the B operands in the xf registers, a chunk of Wp goes through the W1 ring, a chain starts from the bias of its columns).  The
    chunk's LDS row 16 j + 4 q + r holds Wp row 32 c + 8 q + 4 j + r (the permutation is applied to the DMA source offset), so that the
    accumulator of hidden-style tile j is exactly oacc[mt][2 c + j]: lane (fr, quad) holds the columns 32 c + 8 quad + 4 j + r.
    The last two chunks' refills already bring the MLP's W1(0) / W1(1)."""
    def wp_items(chunk):          # Wp chunk `chunk` -> W1 slot chunk & 1
        items = []
        for qq in range(8):
            src = chunk * SLOT + (8 * (qq & 3) + 4 * (qq >> 2)) * 1024
            items.append(([f"s_add_u32 m0, s{S_LB1}, {qq * 4096}", f"s_add_u32 s{S_T2}, {SW}, {src}"],
                          [f"buffer_load_dwordx4 {V1(qq & 3)}, {WPR}, s{S_T2} offen lds"]))
        return items

    def w1_items(chunk):          # the MLP's W1 chunk (natural row order) -> W1 slot chunk & 1
        items = []
        for qq in range(8):
            items.append(([f"s_add_u32 m0, s{S_LB1}, {qq * 4096}", f"s_add_u32 s{S_T2}, {SW}, {chunk * SLOT + qq * 4096}"],
                          [f"buffer_load_dwordx4 {V1(qq & 3)}, {W1R}, s{S_T2} offen lds"]))
        return items

    E.c("---- PROJ: the attention output rows -> operand registers; Wp(0 .. 2) -> ring slots 0 .. 2 (during the projection all FOUR 32-KiB")
    E.c("slots of the two rings hold Wp chunks: chunk c in slot c & 3, requested three chunks ahead, counted waits)")
    E.vq = []
    E.i(f"s_mov_b32 s{S_MT1B}, {16 * 1024}")
    for mt in range(2):
        for ks in range(16):
            so = "0" if mt == 0 else f"s{S_MT1B}"
            E.vm(("att", mt, ks), f"buffer_load_dwordx4 {XF(mt, ks)}, {XBOFF}, {ATTR}, {so} offen offset:{64 * ks}")

    def issue(chunk):             # ring refill `chunk` (16, 17: the MLP's W1(0), W1(1); >= 18: nothing) -> slot chunk & 3
        if chunk >= 18:
            return []
        items = wp_items(chunk) if chunk < 16 else w1_items(chunk - 16)
        out = []
        for qq, (pre, load) in enumerate(items):
            pre = [pre[0].replace(f"s{S_LB1}, {qq * 4096}", f"{SLW}, {(chunk & 3) * SLOT + qq * 4096}"), pre[1]]
            out.append((pre, load, ("w", chunk, qq)))
        return out

    for chunk in range(3):
        for pre, load, tag in issue(chunk):
            for l in pre:
                E.i(l)
            E.vm(tag, load[0])
    E.i("s_waitcnt lgkmcnt(0)")
    cur = 0                       # slot the fragment bases point at
    for c in range(16):
        E.c(f"projection chunk {c}: columns {32 * c} .. {32 * c + 31}")
        E.vm_wait(("w", c, 7))    # this wave's pieces of chunk c (and everything older: the attention rows); younger requests stay in flight
        E.i("s_barrier")
        for j in range(2):
            E.ds_read(("b", j), BZ(j), VECP, 6144 + 128 * c + 16 * j)
        for g in range(32, 32 + PF):
            a, off = frag_addr(g)
            E.ds_read(("f", g), WQ(g % PF), a, off)
        # slot (c + 3) & 3 = (c - 1) & 3 was read by chunk c - 1: free behind the barrier above
        items = issue(c + 3)
        for g in range(32, 64):
            k = g - 32
            if k % 2 == 0 and k // 2 < len(items):
                pre, load, tag = items[k // 2]
                group(E, g, 64, between=pre)
                E.vm(tag, load[0])
            else:
                group(E, g, 64)
        assert not E.q
        E.c("MFMA D -> vector reader")
        E.i("s_nop 7")
        E.i("s_nop 3")
        for mt in range(2):
            for j in range(2):
                for r in range(4):
                    E.i(f"v_accvgpr_write_b32 {ACCR(mt, 2 * c + j, r)}, v{HACC(mt, j) + r}")
        nxt = (c + 1) & 3
        for a in range(4):
            E.i(f"v_xor_b32 {W1P(a)}, {hex((cur ^ nxt) * SLOT)}, {W1P(a)}")
        cur = nxt
    assert cur == 0
    E.c("every wave is done with the ring: W2(0) -> W2 slot 0 (the MLP's W1(0), W1(1) arrived as chunks 16, 17)")
    E.i("s_barrier")
    E.i(f"s_mov_b32 s{S_SLOT2}, 0")
    E.i(f"s_mov_b32 s{S_SO2}, {SW}")
    E.i(f"s_mov_b32 s{S_SLOT}, 0")
    dma_bases(E)
    dma(E, "w2")
    E.vq = []


def proj_ln(E, gam=8192, bet=10240, agent_scope=False, head=None):
    """x_new = x + LayerNorm(oacc) * gamma + beta for the wave's rows; x_new goes back to memory as fp32 and, rounded to bf16, into the xf
    registers: the 8 consecutive columns a lane holds per column group ARE the B operand of that k-step of a GEMM-1-type product.
    The residual rows stream through a ring of eight quads requested eight quads ahead.
    PROJ (defaults): oacc = att Wp^T + bp, gamma1 / beta1, the residual is x; x1 is what the MLP's own LayerNorm reads back later.
    QKV epilogue (gam / bet of norm2, agent_scope: the residual is the x1 this wave stored earlier in the launch): the registers then
    hold the shadow of the block's OUTPUT -- the operand of the next block's qkv Linear."""
    XOFF = T
    MEAN = [T + 2, T + 4]
    RSTD = [T + 6, T + 8]
    S, TMP = T + 10, T + 11
    P0, P1, W = T + 16, T + 18, T + 20
    RING = 8
    def XW(n): return 224 + 4 * (n % RING)        # v224..v255 (the GELU's registers: its constants are set up behind this phase)
    E.c("---- LayerNorm + residual, results to memory (fp32) and to the operand registers (bf16)")
    E.i(f"v_lshlrev_b32 v{XOFF}, 1, {XBOFF}")
    E.i(f"s_mov_b32 s{S_MT1X}, {16 * 2048}")
    def load(n):
        mt, jo = n // 32, n % 32
        so = "0" if mt == 0 else f"s{S_MT1X}"
        E.vm(("x", n), f"buffer_load_dwordx4 {vq(XW(n))}, v{XOFF}, {XR}, {so} offen offset:{col_off(jo)}" + (" sc1" if agent_scope else ""))
    if head is None:
        E.vq = []
    else:
        head()
    for n in range(RING):
        load(n)
    ln_stats(E, MEAN, RSTD, S, TMP, P0, P1, W)
    for mt in range(2):
        for pp in range(16):
            base = T + 24 + 24 * (pp & 1)         # register set of this column group: Y0, Y1, G0, B0, G1, B1
            Y = [base, base + 4]
            G = [base + 8, base + 16]
            B = [base + 12, base + 20]
            for t in range(2):
                jo = 2 * pp + t
                E.ds_read(("g", jo), G[t], VECP, gam + col_off(jo))
                E.ds_read(("e", jo), B[t], VECP, bet + col_off(jo))
            for t in range(2):
                jo = 2 * pp + t
                n = 32 * mt + jo
                for r in range(4):
                    E.i(f"v_accvgpr_read_b32 v{Y[t] + r}, {ACCR(mt, jo, r)}")
                for h in (0, 2):
                    E.i(f"v_pk_add_f32 {vp(Y[t] + h)}, {vp(Y[t] + h)}, {vp(MEAN[mt])} op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]")
                for h in (0, 2):
                    E.i(f"v_pk_mul_f32 {vp(Y[t] + h)}, {vp(Y[t] + h)}, {vp(RSTD[mt])} op_sel_hi:[1,0]")
                E.wait(("e", jo))
                for h in (0, 2):
                    E.i(f"v_pk_fma_f32 {vp(Y[t] + h)}, {vp(Y[t] + h)}, {vp(G[t] + h)}, {vp(B[t] + h)}")
                E.vm_wait(("x", n))
                for h in (0, 2):
                    E.i(f"v_pk_add_f32 {vp(Y[t] + h)}, {vp(Y[t] + h)}, {vp(XW(n) + h)}")
                so = "0" if mt == 0 else f"s{S_MT1X}"
                E.vm(("s", n), f"buffer_store_dwordx4 {vq(Y[t])}, v{XOFF}, {XR}, {so} offen offset:{col_off(jo)}")
                if n + RING < 64:
                    load(n + RING)
            xf = 4 * (16 * mt + pp)               # XF(mt, pp)
            for t in range(2):
                E.i(f"v_cvt_pk_bf16_f32 v{xf + 2 * t}, v{Y[t]}, v{Y[t] + 1}")
                E.i(f"v_cvt_pk_bf16_f32 v{xf + 2 * t + 1}, v{Y[t] + 2}, v{Y[t] + 3}")
    assert not E.q


def qkv_phase(E):
    """QKV: the NEXT block's qkv Linear on this block's output rows, which never leave the registers as bf16:
        qkv[rows, 0:1536] = bf16(x_out) Wqkv^T + (q_bias | 0 | v_bias)
    on GEMM 1's machinery, 64 columns (two 32-row chunks A, B of Wqkv = two ring slots) per barrier: 24 double chunks, double chunk s
    in slots 2 (s & 1) and + 1, the next one requested while this one is multiplied (Wqkv's rows permuted at the DMA source so that a
    lane ends with 8 consecutive columns per chunk: one 16-byte store per row tile and chunk).  The 64 MFMA groups of a double chunk
    are ONE stream -- the fragment ring runs across the A / B boundary, B accumulates in a second register set and A's bias + rounding
    + stores sit behind B's first two groups, so the matrix pipe never drains for them.  The accumulation starts from zero and the
    bias is added at the end -- the order of the stand-alone qkv GEMM.  Replaces a launch that read the 67-MB shadow back (which is
    then not written at all).
    A LOOP of 12 passes x 2 double chunks (the ring's period): everything that depends on the pass sits in three scalar registers
    and one vector register that advance once per pass.
    Measured on the way here (tools/micro/qkv512_phase.py, profiles/r05_qkv512_phase.txt): one barrier per 32-column chunk cost
    2 270 cycles per chunk against 1 024 of MFMA issue -- 1 850 of them in the 32 groups themselves, the same groups that take
    1 300 inside the MLP loop's 64-group iterations: what a barrier costs is the four waves' drift plus a cold fragment ring."""
    NS = 24
    STAMP = bool(ABL & 1024)
    PF = 10 if ABL & 8192 else 8                 # (experiment: a deeper fragment ring -- v160..v199)
    SPARSE = bool(ABL & 16384)                   # (experiment: a ring refill every other group of the first 32 instead of every group of the first 16)
    S_QO1, S_QI, S_QB, S_QO0 = 87, 88, 89, 90    # byte offset of the second / first 16-row tile in the qkv rows (+ 256 per pass); passes left; DMA source base
    VB, QOFF = "v249", "v248"                    # bias base (+ 512 per pass); row * 3072 + 16 quad
    def HACC1(mt, j): return 200 + 4 * (2 * mt + j)      # B's accumulators (the GELU's fragment registers)
    def BZ1(j): return 232 + 4 * j                       # B's bias
    KSET = [[224, 228], [240, 244]]                      # rounded results of A / B (v224..v231, v240..v247)

    def perm(chunk, qq):
        return chunk * SLOT + (8 * (qq & 3) + 4 * (qq >> 2)) * 1024

    def wq_items(chunk, rolled=False):
        """LDS-DMA of chunk `chunk` into slot chunk & 3; rolled: requested by the pass before its own (chunk = 4 pass + r, r = 2 .. 5
        relative to the requesting pass): source = S_QB (this wave's rows + 4 pass SLOT) + the rest.  The last pass requests chunks
        48, 49, past Wqkv's end: zeros arrive, nobody reads them -- the scalar offset is part of the descriptor's range check on
        this part: tools/micro/buffer_soffset_range.hip"""
        items = []
        for qq in range(8):
            if rolled:
                src, base = perm(chunk - 4 * ((chunk - 2) >> 2), qq), f"s{S_QB}"
            else:
                src, base = perm(chunk, qq), SW
            items.append(([f"s_add_u32 m0, {SLW}, {(chunk & 3) * SLOT + qq * 4096}", f"s_add_u32 s{S_T2}, {base}, {src}"],
                          f"buffer_load_dwordx4 {V1(qq & 3)}, {WQR}, s{S_T2} offen lds", ("q", chunk, qq)))
        return items

    def head():
        E.c("every wave is past the last GEMM 2: the ring takes the first double chunk of the next block's Wqkv")
        E.vq = []
        E.i("s_barrier")
        for chunk in range(2):
            for pre, load, tag in wq_items(chunk):
                for l in pre:
                    E.i(l)
                E.vm(tag, load)

    def qstamp(k):
        """diagnostic: s[70 + 2k : +1] = s_memtime (an SMEM return: only where the LDS queue is empty)"""
        if STAMP:
            assert not E.q
            E.i(f"s_memtime s[{70 + 2 * k}:{71 + 2 * k}]")
            E.i("s_waitcnt lgkmcnt(0)")

    def finish(c, half):
        """+ bias, round, store the 32 columns of chunk c (r = its position in the pass)"""
        if ABL & 64:
            return
        r = c & 3
        H, B, K = (HACC, BZ, KSET[0]) if half == 0 else (HACC1, BZ1, KSET[1])
        for mt in range(2):
            for j in range(2):
                for h in (0, 2):
                    E.i(f"v_pk_add_f32 {vp(H(mt, j) + h)}, {vp(H(mt, j) + h)}, {vp(B(j) + h)}")
            for j in range(2):
                E.i(f"v_cvt_pk_bf16_f32 v{K[mt] + 2 * j}, v{H(mt, j)}, v{H(mt, j) + 1}")
                E.i(f"v_cvt_pk_bf16_f32 v{K[mt] + 2 * j + 1}, v{H(mt, j) + 2}, v{H(mt, j) + 3}")
            if not (ABL & 4096):    # (diagnostic: the arithmetic without the stores)
                E.vm(("o", c, mt), f"buffer_store_dwordx4 {vq(K[mt])}, {QOFF}, {QR}, s{S_QO0 if mt == 0 else S_QO1} offen offset:{64 * r}")

    def frag(G):
        a, off = frag_addr(32 + (G & 31))
        return a, off + SLOT * (G >> 5)

    def double_chunk(s_):
        """double chunk s_ = chunks 2 s_ (A) and 2 s_ + 1 (B), ring slots 2 (s_ & 1) and + 1 (the fragment bases point at A's)"""
        cA, cB = 2 * s_, 2 * s_ + 1
        E.c(f"qkv double chunk {s_}: columns {64 * s_} .. {64 * s_ + 63}")
        qstamp(0)
        E.vm_wait(("q", cB, 7))
        if ABL & 16 and s_ >= 1:
            E.vm_wait(("o", cB - 2, 1))
        if not (ABL & 32):
            E.i("s_barrier")
        qstamp(1)
        for half, B in ((0, BZ), (1, BZ1)):
            for j in range(2):
                E.ds_read(("b", half, j), B(j), VB, 12288 + 128 * ((cA + half) & 3) + 16 * j)
        if not (ABL & 128):
            for G in range(PF):
                a, off = frag(G)
                E.ds_read(("f", G), WQ(G % PF), a, off)
        items = [] if ABL & 16 else wq_items(cA + 2, True) + wq_items(cB + 2, True)
        for G in range(64):
            half, e = G >> 5, G & 31
            j, ks = e & 1, e >> 1
            H = HACC if half == 0 else HACC1
            E.wait(("f", G))
            w = vq(WQ(G % PF))
            k = G // 2 if SPARSE and G % 2 == 0 else (G if not SPARSE else 99)
            pre, load, tag = items[k] if k < len(items) else ((), None, None)      # all sixteen requests early: they have the rest of this double chunk to land
            for mt in range(2):
                if not (ABL & 8):
                    E.i(f"v_mfma_f32_16x16x32_bf16 {vq(H(mt, j))}, {w}, {XF(mt, ks)}, {'0' if ks == 0 else vq(H(mt, j))}")
                if mt == 0:
                    for l in pre:
                        E.i(l)
            if G + PF < 64 and not (ABL & 128):
                a, off = frag(G + PF)
                E.ds_read(("f", G + PF), WQ(G % PF), a, off)
            if load:
                E.vm(tag, load)
            if G == 33:
                E.c("A's results (its last MFMA was issued 4 MFMAs = 64 cycles ago): + bias, round, store")
                finish(cA, 0)
        assert not E.q or all(t[0] == "b" for t in E.q)
        E.wait_all()
        E.c("MFMA D -> vector reader; B's results")
        E.i("s_nop 7")
        E.i("s_nop 3")
        qstamp(2)
        finish(cB, 1)
        for a in range(4):
            E.i(f"v_xor_b32 {W1P(a)}, {hex(2 * SLOT)}, {W1P(a)}")
        qstamp(3)
        for k in range(3):
            if STAMP:
                E.i(f"s_sub_u32 s{S_T}, s{70 + 2 * (k + 1)}, s{70 + 2 * k}")
                E.i(f"s_add_u32 s{80 + k}, s{80 + k}, s{S_T}")

    E.c("---- epilogue of the QKV variant: MFMA D -> v_accvgpr_read")
    E.i("s_nop 15")
    proj_ln(E, gam=2048, bet=4096, agent_scope=PROJ, head=head)
    E.i(f"s_mov_b32 s{S_QO0}, 0")
    E.i(f"s_mov_b32 s{S_QO1}, {16 * 3072}")
    E.i(f"s_mov_b32 s{S_QB}, {SW}")
    E.c("row * 3072 + 16 quad = 3 xboff - 2 (16 quad), xboff = row * 1024 + 16 quad")
    E.i(f"v_and_b32 {VB}, 0x3ff, {XBOFF}")
    E.i(f"v_lshl_add_u32 {QOFF}, {XBOFF}, 1, {XBOFF}")
    E.i(f"v_lshlrev_b32 {VB}, 1, {VB}")
    E.i(f"v_sub_u32 {QOFF}, {QOFF}, {VB}")
    E.i(f"v_mov_b32 {VB}, {VECP}")
    if STAMP:
        E.i("s_memtime s[92:93]")
        for k in range(4):
            E.i(f"s_mov_b32 s{80 + k}, 0")
    # the pass's text in the steady state: generate pass 0 (thrown away), pass 1 (kept), pass 2 (must be the same text).  The counted
    # vmcnt waits of the steady state are right for pass 0 as well: there the operations younger than a wait's target are the
    # steady state's plus the LayerNorm's loads and stores, all of them issued after the target -- the wait only asks for more.
    texts = []
    for it in range(3):
        mark = len(E.lines)
        for h in range(2):
            double_chunk(2 * it + h)
        texts.append(E.lines[mark:])
        del E.lines[mark:]
    strip = lambda t: [l for l in t if not l.startswith(";")]
    assert strip(texts[1]) == strip(texts[2])
    E.i(f"s_mov_b32 s{S_QI}, {NS // 2}")
    E.i("L_qkv%=:")
    E.lines.extend(texts[1])
    E.c("next pass: 4 chunks = 4 ring slots of Wqkv rows, 128 columns of bias and of the qkv rows")
    E.i(f"s_add_u32 s{S_QB}, s{S_QB}, {4 * SLOT}")
    E.i(f"s_add_u32 s{S_QO0}, s{S_QO0}, 256")
    E.i(f"s_add_u32 s{S_QO1}, s{S_QO1}, 256")
    E.i(f"v_add_u32 {VB}, 512, {VB}")
    E.i(f"s_sub_u32 s{S_QI}, s{S_QI}, 1")
    E.i(f"s_cmp_lg_u32 s{S_QI}, 0")
    E.i("s_cbranch_scc1 L_qkv%=")
    if STAMP:
        E.c("diagnostic: (the whole phase | waiting for the LDS-DMA and at the barrier | fragments + MFMAs + A's results | B's results), shader cycles summed over the double chunks, into the first 16 bytes of the lane's qkv row")
        E.i("s_memtime s[94:95]")
        E.i("s_waitcnt lgkmcnt(0)")
        E.i("s_sub_u32 s92, s94, s92")
        E.i("v_mov_b32 v208, s92")
        for k in range(3):
            E.i(f"v_mov_b32 v{209 + k}, s{80 + k}")
        E.i(f"buffer_store_dwordx4 v[208:211], {QOFF}, {QR}, 0 offen")
    E.vq = []


def epilogue(E):
    """x += LayerNorm(oacc) * gamma + beta (oacc already holds the fc2 bias), shadow = bf16(x), for the wave's two 16-row tiles.
    Lane (fr, quad) holds, of row 16 mt + fr, the columns 32 pp + 8 quad + 4 t + r in ACC(mt, 2 pp + t)[r]."""
    XOFF = T                       # v168: row * 2048 + 32 quad
    MEAN = [T + 2, T + 4]          # pairs (value in the low half)
    RSTD = [T + 6, T + 8]
    S, TMP = T + 10, T + 11
    P0, P1, W = T + 16, T + 18, T + 20           # 184:185, 186:187, 188:191
    def XIN(jo): return 4 * jo                    # the xf area: 32 quads
    E.c("---- epilogue")
    E.i("s_nop 15")
    E.i(f"v_lshlrev_b32 v{XOFF}, 1, {XBOFF}")
    E.i(f"s_mov_b32 s{S_MT1X}, {16 * 2048}")
    E.i(f"s_mov_b32 s{S_MT1B}, {16 * 1024}")
    E.c("residual rows of tile 0 (they land under the statistics)")
    for jo in range(32):
        E.i(f"buffer_load_dwordx4 {vq(XIN(jo))}, v{XOFF}, {XR}, 0 offen offset:{col_off(jo)}" + SC1())
    ln_stats(E, MEAN, RSTD, S, TMP, P0, P1, W)
    for mt in range(2):
        E.c(f"tile {mt}: normalise, add the residual rows, store x and the shadow")
        E.i("s_waitcnt vmcnt(0)")
        for pp in range(16):
            base = T + 24 + 28 * (pp & 1)         # register set of this column group: Y0, Y1, K, G0, B0, G1, B1
            Y = [base, base + 4]
            K = base + 8
            G = [base + 12, base + 20]
            B = [base + 16, base + 24]
            for t in range(2):
                jo = 2 * pp + t
                E.ds_read(("g", jo), G[t], VECP, 2048 + col_off(jo))
                E.ds_read(("e", jo), B[t], VECP, 4096 + col_off(jo))
            for t in range(2):
                jo = 2 * pp + t
                for r in range(4):
                    E.i(f"v_accvgpr_read_b32 v{Y[t] + r}, {ACCR(mt, jo, r)}")
                for h in (0, 2):
                    E.i(f"v_pk_add_f32 {vp(Y[t] + h)}, {vp(Y[t] + h)}, {vp(MEAN[mt])} op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]")
                for h in (0, 2):
                    E.i(f"v_pk_mul_f32 {vp(Y[t] + h)}, {vp(Y[t] + h)}, {vp(RSTD[mt])} op_sel_hi:[1,0]")
                E.wait(("e", jo))
                for h in (0, 2):
                    E.i(f"v_pk_fma_f32 {vp(Y[t] + h)}, {vp(Y[t] + h)}, {vp(G[t] + h)}, {vp(B[t] + h)}")
                for h in (0, 2):
                    E.i(f"v_pk_add_f32 {vp(Y[t] + h)}, {vp(Y[t] + h)}, {vp(XIN(jo) + h)}")
                so = "0" if mt == 0 else f"s{S_MT1X}"
                E.i(f"buffer_store_dwordx4 {vq(Y[t])}, v{XOFF}, {XR}, {so} offen offset:{col_off(jo)}")
                if mt == 0:
                    E.c("the same quad of tile 1's residual rows takes the place of the one just used")
                    E.i(f"buffer_load_dwordx4 {vq(XIN(jo))}, v{XOFF}, {XR}, s{S_MT1X} offen offset:{col_off(jo)}" + SC1())
            for t in range(2):
                E.i(f"v_cvt_pk_bf16_f32 v{K + 2 * t}, v{Y[t]}, v{Y[t] + 1}")
                E.i(f"v_cvt_pk_bf16_f32 v{K + 2 * t + 1}, v{Y[t] + 2}, v{Y[t] + 3}")
            so = "0" if mt == 0 else f"s{S_MT1B}"
            E.i(f"buffer_store_dwordx4 {vq(K)}, {XBOFF}, {XBR}, {so} offen offset:{64 * pp}")
        assert not E.q


def program():
    """Schedule of one hidden chunk (iteration i), one workgroup barrier per chunk:
        [bias values of chunk i requested] barrier [first fragments requested]
        GELU(i): ONE uninterrupted run of vector instructions -> hf      (the fragments' LDS round trip lands under it)
        GEMM 2 of chunk i (32 MFMA groups; one LDS-DMA item behind each of the first 16), GEMM 1 of chunk i+1 (32 groups): one run of MFMAs
    reads W2 slot i & 1 and W1 slot (i+1) & 1; refills W2 slot (i+1) & 1 with chunk i+1 and W1 slot i & 1 with chunk i+2.
    Why runs and not an interleave: on this part the matrix and the vector pipe do not overlap, and a vector instruction behind an
    MFMA costs ~16 cycles where it costs 5 behind another vector instruction (tools/micro/single_wave_issue.hip: 64 MFMAs with 32
    packed FMAs between them take 24.3 cycles per MFMA, alone 16.1) -- the first version, with the polynomial's steps between the
    MFMA groups, ran 3 500 cycles per chunk against 2 064 of MFMAs + 650 of vector work."""
    E = Emit()
    E.c("---- set-up: constants, m0 saved (s_nop 4: a scalar operand may be fresh from a v_readfirstlane)")
    E.i("s_nop 4")
    E.i(f"s_mov_b32 s{S_M0}, m0")
    if TIMING:
        for a in range(6):
            E.i(f"s_mov_b32 s{S_ACC + a}, 0")
        stamp(E, 4)      # kernel start
    if STAGGER:
        E.c("start skew: the workgroups of the first round (one per CU) start in four phases, so that the chip-wide bursts of their row")
        E.c("loads / stores (every CU moves 768 KB at the same moment otherwise) fall under the other phases' MFMAs")
        E.i(f"s_lshr_b32 s{S_T}, {BID}, 3")
        E.i(f"s_and_b32 s{S_T}, s{S_T}, 3")
        E.i(f"s_cmp_lt_u32 {BID}, 256")
        E.i(f"s_cselect_b32 s{S_T}, s{S_T}, 0")
        E.i(f"s_mul_i32 s{S_T}, s{S_T}, {STAGGER}")
        E.i("L_skew%=:")
        E.i(f"s_cmp_eq_u32 s{S_T}, 0")
        E.i("s_cbranch_scc1 L_skew_done%=")
        E.i("s_sleep 32")
        E.i(f"s_sub_u32 s{S_T}, s{S_T}, 1")
        E.i("s_branch L_skew%=")
        E.i("L_skew_done%=:")
    E.i(f"s_mov_b32 s{S_NU}, {f32(-GELU_U)}")
    E.i(f"s_mov_b32 s{S_T}, {f32(GELU_U)}")
    E.i(f"v_mov_b32 v{VU}, s{S_T}")
    E.i(f"s_mov_b32 s{S_T}, {f32(GELU_C[GELU_DEG - 1])}")
    E.i(f"v_mov_b32 v{VC7}, s{S_T}")
    E.i(f"v_mov_b32 v{VC7 + 1}, s{S_T}")
    for b, val in [(S_ZS, GELU_ZS), (S_C8, GELU_C[GELU_DEG])] + [(S_C(k), GELU_C[k]) for k in range(GELU_DEG - 1)]:
        E.i(f"s_mov_b32 s{b}, {f32(val)}")
        E.i(f"s_mov_b32 s{b + 1}, {f32(val)}")
    if PROJ:
        proj_gemm(E)
        proj_ln(E)
        E.c("the GELU's vector constants (their registers served as the residual ring above)")
        E.i(f"s_mov_b32 s{S_T}, {f32(GELU_U)}")
        E.i(f"v_mov_b32 v{VU}, s{S_T}")
        E.i(f"s_mov_b32 s{S_T}, {f32(GELU_C[GELU_DEG - 1])}")
        E.i(f"v_mov_b32 v{VC7}, s{S_T}")
        E.i(f"v_mov_b32 v{VC7 + 1}, s{S_T}")
    else:
        E.c("---- W1(0) -> W1 slot 0, W1(1) -> W1 slot 1, W2(0) -> W2 slot 0")
        E.i(f"s_mov_b32 s{S_SLOT2}, 0")
        E.i(f"s_mov_b32 s{S_SO2}, {SW}")
        for ch in range(2):
            E.i(f"s_mov_b32 s{S_SLOT}, {ch * SLOT}")
            E.i(f"s_add_u32 s{S_SO1}, {SW}, {ch * SLOT}")
            dma_bases(E)
            dma(E, "w1")
        dma(E, "w2")
    E.c("iteration 0 refills W1 slot 0 with chunk 2 and W2 slot 1 with chunk 1")
    E.i(f"s_mov_b32 s{S_SLOT}, 0")
    E.i(f"s_mov_b32 s{S_SLOT2}, {SLOT}")
    E.i(f"s_add_u32 s{S_SO1}, {SW}, {2 * SLOT}")
    E.i(f"s_add_u32 s{S_SO2}, {SW}, {SLOT}")
    E.i(f"s_mov_b32 s{S_I}, 0")
    if not PROJ:
        E.c("---- this wave's rows of xb as B operands of GEMM 1: xf[mt][ks] = xb[row0 + 16 mt + fr][32 ks + 8 quad .. + 7]")
        E.i(f"s_mov_b32 s{S_MT1B}, {16 * 1024}")
        for mt in range(2):
            for ks in range(16):
                so = "0" if mt == 0 else f"s{S_MT1B}"
                E.i(f"buffer_load_dwordx4 {XF(mt, ks)}, {XBOFF}, {XBR}, {so} offen offset:{64 * ks}")
    E.c("the compiler's LDS stores of the bias rows (lgkmcnt) and this wave's DMA pieces and rows (vmcnt), then everybody's")
    E.i("s_waitcnt vmcnt(0) lgkmcnt(0)")
    E.i("s_barrier")
    E.c("---- oacc = the fc2 bias of its columns (the MFMAs accumulate on top of it)")
    for jo0 in range(0, 32, 8):
        for jo in range(jo0, jo0 + 8):
            E.ds_read(("i", jo), WQ(0) + 4 * (jo - jo0), VECP, col_off(jo))
        for jo in range(jo0, jo0 + 8):
            E.wait(("i", jo))
            for mt in range(2):
                for r in range(4):
                    E.i(f"v_accvgpr_write_b32 {ACCR(mt, jo, r)}, v{WQ(0) + 4 * (jo - jo0) + r}")
    E.c("---- GEMM 1 of chunk 0 (W1 slot 0); its chains start from the fc1 bias of the chunk's hidden units")
    E.ds_read(("b", 0), BZ(0), B1P, 0)
    E.ds_read(("b", 1), BZ(1), B1P, 64)
    E.i(f"v_add_u32 {B1P}, 0x80, {B1P}")
    for g in range(32, 32 + PF):
        a, off = frag_addr(g)
        E.ds_read(("f", g), WQ(g % PF), a, off)
    for g in range(32, 64):
        group(E, g, 64)
    assert not E.q
    E.i("s_nop 7")
    E.c("iteration 0 reads W2 slot 0 and W1 slot 1")
    for a in range(4):
        E.i(f"v_xor_b32 {W1P(a)}, 0x8000, {W1P(a)}")

    E.i("L_top%=:")
    stamp(E, 0)
    E.i("s_waitcnt vmcnt(0)")
    stamp(E, 1)
    E.i("s_barrier")
    stamp(E, 2)
    if TIMING:
        lap(E, 0, 1, 0)      # waiting for this wave's DMA pieces
        lap(E, 1, 2, 1)      # waiting for the other waves
    for g in range(PF):
        if ABL & 4:
            break
        a, off = frag_addr(g)
        E.ds_read(("f", g), WQ(g % PF), a, off)
    dma_bases(E)
    for mt in range(2):
        gelu(E, mt, lambda step: None)
    E.c("the fc1 bias of chunk i + 1: GEMM 1's chains of this iteration start from it (GELU(i) above was the last reader of chunk i's)")
    E.ds_read(("b", 0), BZ(0), B1P, 0)
    E.ds_read(("b", 1), BZ(1), B1P, 64)
    E.i(f"v_add_u32 {B1P}, 0x80, {B1P}")
    E.c("(vector write -> MFMA operand: the two LDS requests and the add above are the wait states)")
    items = [] if ABL & 2 else dma_items("w2") + dma_items("w1")
    if ABL & 4:
        E.wait_all()
    for g in range(32):
        pre, load = items[g // 2] if g % 2 == 0 and g // 2 < len(items) else ((), ())
        group(E, g, 64, between=pre, after=load)
    E.i(f"s_cmp_eq_u32 s{S_I}, {NCH - 1}")
    E.i("s_cbranch_scc1 L_done%=")
    for g in range(32, 64):
        group(E, g, 64)
    assert not E.q
    E.i("s_nop 7")
    for a in range(4):
        E.i(f"v_xor_b32 {W1P(a)}, 0x8000, {W1P(a)}")
    E.i(f"v_xor_b32 {W2P}, 0x8000, {W2P}")
    E.i(f"s_xor_b32 s{S_SLOT}, s{S_SLOT}, 0x8000")
    E.i(f"s_xor_b32 s{S_SLOT2}, s{S_SLOT2}, 0x8000")
    E.i(f"s_add_u32 s{S_SO1}, s{S_SO1}, {SLOT}")
    E.i(f"s_add_u32 s{S_SO2}, s{S_SO2}, {SLOT}")
    E.i(f"s_add_u32 s{S_I}, s{S_I}, 1")
    stamp(E, 3)
    lap(E, 2, 3, 2)          # the iteration's work behind the barrier
    E.i("s_branch L_top%=")

    E.i("L_done%=:")
    E.c("(the last iteration requested the first fragments of a GEMM 1 that does not exist)")
    E.i("s_waitcnt lgkmcnt(0)")
    E.q = []
    stamp(E, 3)
    if QKV:
        qkv_phase(E)
    else:
        epilogue(E)
    E.i("s_waitcnt vmcnt(0)")
    if TIMING:
        stamp(E, 0)
        lap(E, 3, 3, 4)      # kernel start -> epilogue start
        lap(E, 4, 0, 3)      # epilogue
        for a in range(5):
            E.i(f"v_mov_b32 v{T}, s{S_ACC + a}")
            E.i(f"buffer_store_dword v{T}, off, {DBGR}, {DBGOFF} offset:{4 * a}")
        E.i("s_waitcnt vmcnt(0)")
    E.i(f"s_mov_b32 m0, s{S_M0}")
    return E.lines


# (PF, STAGGER, TIMING, ABL, PROJ): fragments requested PF groups ahead; start skew of the first round's workgroups in units of s_sleep 32 (2 048
# cycles) per phase (four phases by (blockIdx >> 3) & 3).
# Variant 0 is what launch_swin_mlp512 runs, variant 1 launch_swin_proj_mlp512; the others are reachable through VSC_SWIN_MLP_ABL=<index>.
VARIANTS = [(8, 0, False, 0, False, False), (8, 0, False, 0, True, False), (8, 0, False, 1, False, False), (8, 0, False, 2, False, False), (8, 0, False, 4, False, False),
            (8, 0, True, 0, False, False), (8, 0, False, 8, False, False), (8, 0, False, 14, False, False), (8, 0, False, 15, False, False), (8, 0, False, 0, True, True)]


def main():
    global PF, STAGGER, TIMING, ABL, PROJ, QKV
    import os
    if os.environ.get("VSC_GEN_QKV_ABL"):   # diagnostic builds (make EXTRA=-DVSC_MLP_ABLATION): the QKV phase's ablations as variants 10 .. 13
        VARIANTS.extend((8, 0, False, a, True, True) for a in (16, 32, 64, 128, 1024, 4096, 8192, 16384, 8192 + 16384))
    out = ["// GENERATED by gen_mlp512_loop.py -- do not edit.  One asm statement per variant: the body of swin_mlp512_kernel<V>."]
    for k, (PF, STAGGER, TIMING, ABL, PROJ, QKV) in enumerate(VARIANTS):
        lines = program()
        out.append(f"// variant {k}: PF = {PF}, STAGGER = {STAGGER}" + (", cycle counters per wave" if TIMING else "") + (f", ablation {ABL} (wrong results)" if ABL else "") + (", PROJ: attention projection + LayerNorm + residual in front" if PROJ else "") + (", QKV: the next block's qkv Linear behind" if QKV else ""))
        out.append(f"#define VSC_MLP512_LOOP_ASM_{k} \\")
        for l in lines:
            if l.startswith(";"):
                continue
            # the operand type is a property of the build (common.h VSC_LP_ASM = "bf16" | "f16"): same instruction classes, same waits
            l = l.replace("v_mfma_f32_16x16x32_bf16", 'v_mfma_f32_16x16x32_" VSC_LP_ASM "').replace("v_cvt_pk_bf16_f32", 'v_cvt_pk_" VSC_LP_ASM "_f32')
            out.append(f'    "{l}\\n" \\')
        out.append('    ""')
    out.append(f"#define VSC_MLP512_VARIANTS {len(VARIANTS)}")
    io = ", ".join([f'[w1p{a}] "+v"(w1p[{a}])' for a in range(4)] + ['[w2p] "+v"(w2p)', '[b1p] "+v"(b1p)'])
    ins = ", ".join([f'[v1{a}] "v"(v1[{a}])' for a in range(4)] +
                    ['[v2] "v"(v2)', '[vecp] "v"(vecp)', '[xboff] "v"(xboff)', '[bp16] "v"(bp16)', '[bp32] "v"(bp32)',
                     '[w1r] "s"(w1r)', '[w2r] "s"(w2r)', '[xr] "s"(xr)', '[xbr] "s"(xbr)', '[swave] "s"(swave)', '[sldsw] "s"(sldsw)', '[eps] "s"(eps)', '[dbgr] "s"(dbgr)', '[dbgoff] "s"(dbgoff)', '[bid] "s"(bid)', '[attr] "s"(attr)', '[wpr] "s"(wpr)'])
    clob = ", ".join([f'"v{r}"' for r in range(0, 128)] + [f'"v{r}"' for r in range(T, 256)] + [f'"a{r}"' for r in range(256)] +
                     [f'"s{r}"' for r in range(40, 100)] + ['"scc"', '"memory"'])
    out.append(f"#define VSC_MLP512_LOOP_OUTS {io}")
    out.append(f"#define VSC_MLP512_LOOP_INS {ins}")
    out.append(f"#define VSC_MLP512_LOOP_CLOBBERS {clob}")
    sys.stdout.write("\n".join(out) + "\n")


if __name__ == "__main__":
    main()
