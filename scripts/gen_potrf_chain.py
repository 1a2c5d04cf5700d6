#!/usr/bin/env python3
"""Generates ekf_vio_amd/csrc/potrf_chain.inc: the hand-scheduled instruction stream that factors one
64 x 16 panel held one matrix row per lane (potrf64_lds, phase F, chol.hip).

Why generated assembly: the phase is one dependent chain per column (clamp -> readlane -> rsq -> scale ->
readlane -> fma), and gfx950 wants wait states between its links (1 after a VALU write before v_readlane,
2 after v_readlane before a VALU reads the SGPR, 1 after v_rsq).  The compiler fills them with s_nop; here they
carry the rank-1 updates of the previous column instead (one v_fmac_f32_dpp each, the multipliers broadcast inside
every 16-lane row from a ds_bpermute copy of the diagonal block's column), so a column costs 7 + (15 - k)
issue slots.  Per element this is the same fma sequence, in the same order, as the plain formulation
(potrf64_lds<0>): the results are bit-identical (scripts/potrf_stamps.py prints a hash of L and the inverses).

Usage: python scripts/gen_potrf_chain.py > ekf_vio_amd/csrc/potrf_chain.inc
"""
import sys


def gen(nowait=False, nofill=False):
    out = []

    def emit(s):
        out.append(s)

    for k in range(16):
        ak = "%%[a%d]" % k
        queue = list(range(k + 1, 16)) if (k >= 1 and not nofill) else []   # pending DPP updates of column k-1
        dprev = "%%[d%d]" % ((k - 1) % 2)
        aprev = "%%[a%d]" % (k - 1)

        def fill(n):
            took = 0
            while took < n and queue:
                j = queue.pop(0)
                emit("v_fmac_f32_dpp %%[a%d], -%s, %s row_newbcast:%d row_mask:0xf bank_mask:0xf" % (j, dprev, aprev, j))
                took += 1
            if took < n:
                emit("s_nop %d" % (n - took - 1))

        emit("; column %d" % k)
        emit("v_max_f32 %%[t], 0x1e3ce508, %s" % ak)          # clamp: max(a_kk, 1e-20f)
        if queue and not nowait:
            emit("s_waitcnt lgkmcnt(0)")                      # the ds_bpermute of column k-1
        fill(1)
        emit("v_readlane_b32 %%[s], %%[t], %%[c0]+%d" % k)
        fill(2)
        emit("v_rsq_f32 %[t], %[s]")
        fill(1)
        emit("v_mul_f32 %s, %s, %%[t]" % (ak, ak))
        if k <= 14:
            if k <= 13:
                emit("ds_bpermute_b32 %%[d%d], %%[addr], %s" % (k % 2, ak))
            else:
                fill(1)
            emit("v_readlane_b32 %%[s], %s, %%[c0]+%d" % (ak, k + 1))
            fill(2)
            emit("v_fma_f32 %%[a%d], -%s, %%[s], %%[a%d]" % (k + 1, ak, k + 1))
            while queue:
                fill(1)
        assert not queue
    return out


def gen_ls():
    """The stream of gen() with the panel's LDS traffic inside it: the sixteen column loads are issued at the top and
    the chain starts as soon as the first two have landed; column k is written back right after its scale (every lane
    to its own address `sb`, advanced by the per-lane stride `ss` after each column: a matrix column for the lanes of
    the panel, a row of the 16x16 inverse for the identity lanes), so nothing but the last write's latency remains
    behind the chain.  The strict upper triangle of the diagonal block is stored as it comes out of the sweep (stale
    values): nothing in LDS reads it and the global stores drop it."""
    out = []
    emit = out.append
    for j in range(16):
        emit("ds_read_b32 %%[a%d], %%[lb] offset:%%[p4]*%d" % (j, j))      # p4 = 4 * PLD: the byte stride of a tile column
    for k in range(16):
        ak = "%%[a%d]" % k
        queue = list(range(k + 1, 16)) if k >= 1 else []
        dprev = "%%[d%d]" % ((k - 1) % 2)
        aprev = "%%[a%d]" % (k - 1)

        def fill(n):
            took = 0
            while took < n and queue:
                j = queue.pop(0)
                emit("v_fmac_f32_dpp %%[a%d], -%s, %s row_newbcast:%d row_mask:0xf bank_mask:0xf" % (j, dprev, aprev, j))
                took += 1
            if took < n:
                emit("s_nop %d" % (n - took - 1))

        emit("; column %d" % k)
        if k == 0:
            emit("s_waitcnt lgkmcnt(14)")                     # a0 and a1 have landed
        emit("v_max_f32 %%[t], 0x1e3ce508, %s" % ak)
        if k == 1:
            emit("s_waitcnt lgkmcnt(0)")                      # the other columns, column 0's write and bpermute
        elif queue:
            emit("s_waitcnt lgkmcnt(0)")
        fill(1)
        emit("v_readlane_b32 %%[s], %%[t], %%[c0]+%d" % k)
        fill(2)
        emit("v_rsq_f32 %[t], %[s]")
        fill(1)
        emit("v_mul_f32 %s, %s, %%[t]" % (ak, ak))
        emit("ds_write_b32 %%[sb], %s" % ak)
        if k <= 14:
            if k <= 13:
                emit("ds_bpermute_b32 %%[d%d], %%[addr], %s" % (k % 2, ak))
            emit("v_readlane_b32 %%[s], %s, %%[c0]+%d" % (ak, k + 1))
            emit("v_add_u32 %[sb], %[ss], %[sb]")
            fill(1)
            emit("v_fma_f32 %%[a%d], -%s, %%[s], %%[a%d]" % (k + 1, ak, k + 1))
            while queue:
                fill(1)
        assert not queue
    emit("s_waitcnt lgkmcnt(0)")                              # the compiler does not see these LDS writes
    return out


def gen_ls3(slots=(1, 2, 1)):
    """gen_ls() rescheduled so that no LDS latency sits on the pivot chain.  Column k's update reaches column k+1 AND
    column k+2 through SGPRs (two v_readlane + two v_fma); only the columns from k+3 on take theirs from the ds_bpermute
    copy (one v_fmac_f32_dpp each).  Those fills go, earliest deadline first, into the chain's wait states of LATER columns:
    a wait state of column c only takes fills of columns <= c-2, whose copy has long landed; the one fill per column that
    cannot wait that long -- (c-1, c+2), due before column c's own SGPR-path update of a[c+2] -- is issued just before that
    update, behind a counted s_waitcnt (LDS operations return in order: column c's ds_write and ds_bpermute stay in
    flight).  Per element the updates are still applied in ascending column order, so the results are bit-identical to
    the plain formulation.  One copy register per column (d0..d12): a column's fills spread over all later columns.
    slots: fills placed behind v_max / the first v_readlane / v_rsq (the minimum is the hazard distance 1 / 2 / 1)."""
    out = []
    emit = out.append
    lds_ops = 0            # LDS operations issued so far (in-order return)
    bperm_seq = {}         # column -> sequence number of its ds_bpermute
    waited_upto = [0]      # every LDS op with sequence number <= this has been waited for
    pending = []           # fills (deadline column, k, j), not yet issued

    def wait_for(seq):
        if waited_upto[0] >= seq:
            return
        emit("s_waitcnt lgkmcnt(%d)" % (lds_ops - seq))
        waited_upto[0] = seq

    for j in range(16):
        emit("ds_read_b32 %%[a%d], %%[lb] offset:%%[p4]*%d" % (j, j))
        lds_ops += 1

    def issue(k, j):
        wait_for(bperm_seq[k])
        emit("v_fmac_f32_dpp %%[a%d], -%%[d%d], %%[a%d] row_newbcast:%d row_mask:0xf bank_mask:0xf" % (j, k, k, j))

    def fill(n, nmin, col):
        """up to n fills of columns <= col-2 (earliest deadline first); at least nmin issue slots are taken (s_nop pads)"""
        took = 0
        pending.sort()
        i = 0
        while took < n and i < len(pending):
            dl, k, j = pending[i]
            # per element the fills must stay in ascending k: an older column's fill of the same a[j] is always earlier
            # in this order (smaller k, same deadline), so skipping only ever skips whole (k > col-2) columns
            if k <= col - 2:
                pending.pop(i)
                issue(k, j)
                took += 1
            else:
                i += 1
        if took < nmin:
            emit("s_nop %d" % (nmin - took - 1))

    def flush_due(col):
        pending.sort()
        while pending and pending[0][0] <= col:
            dl, k, j = pending.pop(0)
            issue(k, j)

    for k in range(16):
        ak = "%%[a%d]" % k
        emit("; column %d" % k)
        if k == 0:
            wait_for(3)                                        # a0, a1, a2 have landed
        elif k == 1:
            wait_for(16)                                       # every column of the panel
        emit("v_max_f32 %%[t], 0x1e3ce508, %s" % ak)
        fill(slots[0], 1, k)
        emit("v_readlane_b32 %%[s], %%[t], %%[c0]+%d" % k)
        fill(slots[1], 2, k)
        emit("v_rsq_f32 %[t], %[s]")
        fill(slots[2], 1, k)
        emit("v_mul_f32 %s, %s, %%[t]" % (ak, ak))
        emit("ds_write_b32 %%[sb], %s" % ak)
        lds_ops += 1
        if k <= 12:
            emit("ds_bpermute_b32 %%[d%d], %%[addr], %s" % (k, ak))    # one register per column: its fills spread far
            lds_ops += 1
            bperm_seq[k] = lds_ops
            for j in range(k + 3, 16):
                pending.append((j - 2, k, j))                  # must precede column j-2's SGPR-path update of a[j]
        if k <= 14:
            emit("v_readlane_b32 %%[s], %s, %%[c0]+%d" % (ak, k + 1))
            if k <= 13:
                emit("v_readlane_b32 %%[s2], %s, %%[c0]+%d" % (ak, k + 2))
            else:
                fill(1, 1, k)
            emit("v_add_u32 %[sb], %[ss], %[sb]")
            emit("v_fma_f32 %%[a%d], -%s, %%[s], %%[a%d]" % (k + 1, ak, k + 1))
            if k <= 13:
                flush_due(k)                                   # every older update of a[k+2] first: ascending order per element
                emit("v_fma_f32 %%[a%d], -%s, %%[s2], %%[a%d]" % (k + 2, ak, k + 2))
    assert not pending, pending
    emit("s_waitcnt lgkmcnt(0)")                              # the compiler does not see these LDS writes
    return out


def gen_split_diag():
    """Round 5: the panel split over two wavefronts.  THIS stream (wavefront 0) factors only the panel's 16 x 16 diagonal
    block, held with row r of the block in lane r of EVERY 16-lane row (four copies): every broadcast the elimination needs
    -- the pivot, the multipliers -- is then a DPP row_newbcast of the operand itself: no v_readlane, no SGPR wait states, no
    ds_bpermute copy.  Per column k: v_rsq_f32_dpp (pivot broadcast inside the instruction), scale, then ONE LDS write publishes
    the column and its reciprocal root together: the second 16-lane row's copy of the column is overwritten with t_k first
    (v_mov_b32_dpp row_bcast:15 row_mask:2 -- that row computes garbage from then on; rows 0, 2, 3 stay exact copies), and all
    64 lanes store to the exchange buffer, 256 bytes per column: [L_k | t_k x 16 | L_k | L_k].  A non-zero t_k is the flag.
    (Measured, scripts/ubench: an LDS write costs the issuing wavefront 5-10 cycles, two in a row 21, a read under 2;
    the first version of this stream, with two writes per column, spent a third of its time on them.)  Then
    L[j] -= L[k][j] * L[k] for j > k, one v_fmac_f32_dpp each.  The other wavefront (gen_split_x) applies the published
    columns to all 64 rows of the panel.  Scheduling: the chain instructions of column k need every update of L[k] done (the
    last one two issue slots back: VALU write -> DPP read); all other updates are issued earliest-target-first in the chain's
    wait states, so early columns come out as early as the issue rate allows.  Per element the updates are applied in
    ascending source column: bit-identical to the plain formulation.  No clamp: a non-positive pivot gives NaN / inf, which
    the caller's diagonal check reports exactly where the clamped streams report it (potrf64_lds_split)."""
    out = []
    emit = out.append
    lds_ops = [0]
    load_seq = {}
    waited = [0]
    hist = []                      # register written by each VALU instruction issued, newest last (None = not a VALU write)

    def note(reg):
        hist.append(reg)

    def gap_ok(reg):               # VALU write -> DPP read of the same VGPR needs two instructions in between
        return reg not in hist[-2:]

    def wait_for(seq):
        if waited[0] >= seq:
            return
        emit("s_waitcnt lgkmcnt(%d)" % min(15, lds_ops[0] - seq))
        waited[0] = seq

    def need_loaded(j):
        wait_for(load_seq[j])

    for j in range(16):
        emit("ds_read_b32 %%[l%d], %%[lbd] offset:%%[p4]*%d" % (j, j))
        lds_ops[0] += 1
        load_seq[j] = lds_ops[0]

    pending = []                   # (target j, source k): source k's column scaled, (k-1, j) issued
    done_src = {j: 0 for j in range(16)}   # number of sources already applied to L[j] (sources 0 .. done_src-1)

    def ready(j, k):
        return done_src[j] == k and gap_ok("l%d" % k)

    def issue(j, k):
        need_loaded(j)
        emit("v_fmac_f32_dpp %%[l%d], -%%[l%d], %%[l%d] row_newbcast:%d row_mask:0xf bank_mask:0xf" % (j, k, k, j))
        note("l%d" % j)
        done_src[j] = k + 1
        pending.remove((j, k))

    def filler(exclude_target=None):
        """one issue slot: the pending update with the smallest target (then source) that may issue here; else s_nop"""
        for (j, k) in sorted(pending):
            if j == exclude_target:
                continue
            if ready(j, k):
                issue(j, k)
                return True
        emit("s_nop 0")
        note(None)
        return False

    for k in range(16):
        emit("; column %d" % k)
        # every update of L[k] (sources 0 .. k-1) first, then two slots of distance to the DPP read
        while done_src[k] < k:
            src = done_src[k]
            if (k, src) in pending and ready(k, src):
                issue(k, src)
            else:
                filler(exclude_target=k)
        while not gap_ok("l%d" % k):
            filler(exclude_target=k)
        need_loaded(k)
        emit("v_rsq_f32_dpp %%[t], %%[l%d] row_newbcast:%d row_mask:0xf bank_mask:0xf" % (k, k))
        note("t")
        filler()                                               # v_rsq (transcendental) -> use: one wait state
        emit("v_mul_f32 %%[l%d], %%[l%d], %%[t]" % (k, k))
        note("l%d" % k)
        # (t was written two slots back: the filler and the v_mul)
        emit("v_mov_b32_dpp %%[l%d], %%[t] row_bcast:15 row_mask:0x2 bank_mask:0xf" % k)
        note("l%d" % k)
        filler()                                               # the store's data not straight out of the VALU
        emit("ds_write_b32 %%[exw], %%[l%d] offset:%d" % (k, 256 * k))
        note(None)
        lds_ops[0] += 1
        for j in range(k + 1, 16):
            pending.append((j, k))
    while pending:
        filler()
    emit("s_waitcnt lgkmcnt(0)")
    return out


SPLIT_X_RING = 4


def gen_split_x(imm_offsets=True):
    """The other half of gen_split_diag(): wavefront 1 holds the panel one matrix row per lane (x0..x15 = the panel's sixteen
    columns: the diagonal block's own rows, which it factors redundantly, the rows below, and in the lanes of the identity rows
    the rows of I, which become the rows of the diagonal block's inverse) and applies column k of the factored block as soon as
    wavefront 0 has published it: x[k] *= t_k, x[j] -= L[k][j] * x[k].  The multipliers come from the exchange buffer with the
    row index replicated over the four 16-lane rows (one ds_read), so the broadcast is again a DPP row_newbcast of the operand.
    t_k doubles as the flag (zero = not yet).  The (t, L) pairs of the next SPLIT_X_RING - 1 columns are always in flight (an LDS
    read costs under two cycles to issue; waited for at once it costs its 50-100 cycles of latency per column): when this
    wavefront is behind, every pair it looks at is already there and it catches up at its own issue rate; when it is ahead
    it re-reads the pair it needs until the flag is up -- out of line, so that the ready path has no taken branch (a taken
    branch costs the wavefront ~30 cycles of instruction fetch).
    imm_offsets: every lane writes column k at sb + k * 4 * PLD (the identity lanes' inverse rows land in the tile's dead
    block, transposed; the caller copies them to Tinv off the critical path); otherwise sb advances by the per-lane stride ss."""
    out = []
    tail = []
    emit = out.append
    R = SPLIT_X_RING
    lds_ops = [0]
    pair_seq = {}

    def request(c):
        emit("ds_read_b32 %%[t%d], %%[ext] offset:%d" % (c % R, 256 * c))
        emit("ds_read_b32 %%[l%d], %%[exr] offset:%d" % (c % R, 256 * c))
        lds_ops[0] += 2
        pair_seq[c] = lds_ops[0]

    for c in range(R):
        request(c)
    for k in range(16):
        r = k % R
        emit("; column %d" % k)
        emit("s_waitcnt lgkmcnt(%d)" % min(15, lds_ops[0] - pair_seq[k]))
        emit("v_cmp_eq_f32 vcc, 0, %%[t%d]" % r)
        emit("s_cbranch_vccnz .Lpsx_poll%d_%%=" % k)
        emit(".Lpsx_go%d_%%=:" % k)
        tail.append(".Lpsx_poll%d_%%=:" % k)
        tail.append("ds_read_b32 %%[t%d], %%[ext] offset:%d" % (r, 256 * k))
        tail.append("ds_read_b32 %%[l%d], %%[exr] offset:%d" % (r, 256 * k))
        tail.append("s_waitcnt lgkmcnt(0)")
        tail.append("v_cmp_eq_f32 vcc, 0, %%[t%d]" % r)
        tail.append("s_cbranch_vccnz .Lpsx_poll%d_%%=" % k)
        tail.append("s_branch .Lpsx_go%d_%%=" % k)
        emit("v_mul_f32 %%[x%d], %%[x%d], %%[t%d]" % (k, k, r))
        fm = ["v_fmac_f32_dpp %%[x%d], -%%[l%d], %%[x%d] row_newbcast:%d row_mask:0xf bank_mask:0xf" % (j, r, k, j) for j in range(k + 1, 16)]
        for q in fm[:2]:
            emit(q)
        if imm_offsets:
            emit("ds_write_b32 %%[sb], %%[x%d] offset:%%[p4]*%d" % (k, k))   # (two slots behind the v_mul: not straight out of the VALU)
        else:
            emit("ds_write_b32 %%[sb], %%[x%d]" % k)
            emit("v_add_u32 %[sb], %[ss], %[sb]")
        lds_ops[0] += 1
        for q in fm[2:]:
            emit(q)
        if k + R < 16:
            request(k + R)
    emit("s_waitcnt lgkmcnt(0)")
    emit("s_branch .Lpsx_end_%=")
    out.extend(tail)
    emit(".Lpsx_end_%=:")
    return out


def main():
    w = sys.stdout.write
    if len(sys.argv) > 1 and sys.argv[1] == "--experiments":  # timing-only variants (wrong results)
        for name, kw in (("nowait", dict(nowait=True)), ("nofill", dict(nofill=True))):
            w("template <int C0>\n__device__ __forceinline__ void potrf_panel16_chain_%s(float (&a)[16], int diag_lane4) {\n    float d0, d1, t, s;\n    asm volatile(\n" % name)
            for ln in gen(**kw):
                w('        "%s\\n\\t"\n' % ln)
            w("        : " + ", ".join('[a%d] "+v"(a[%d])' % (i, i) for i in range(16)) + ",\n")
            w('          [d0] "=&v"(d0), [d1] "=&v"(d1), [t] "=&v"(t), [s] "=&s"(s)\n')
            w('        : [addr] "v"(diag_lane4), [c0] "n"(C0));\n}\n')
        return
    lines = gen()
    w("// GENERATED by scripts/gen_potrf_chain.py -- do not edit by hand (see that script for the why and the layout).\n")
    w("// Factor phase of one 64 x 16 panel, one matrix row per lane: a[j] = column C0 + j of the panel.\n")
    w("// diag_lane4 = 4 * (C0 + lane % 16): ds_bpermute address of this lane's row of the diagonal block.\n")
    w("template <int C0>\n")
    w("__device__ __forceinline__ void potrf_panel16_chain(float (&a)[16], int diag_lane4) {\n")
    w("    float d0, d1, t, s;\n")
    w("    asm volatile(\n")
    for ln in lines:
        w('        "%s\\n\\t"\n' % ln)
    w("        : " + ", ".join('[a%d] "+v"(a[%d])' % (i, i) for i in range(16)) + ",\n")
    w('          [d0] "=&v"(d0), [d1] "=&v"(d1), [t] "=&v"(t), [s] "=&s"(s)\n')
    w('        : [addr] "v"(diag_lane4), [c0] "n"(C0));\n')
    w("}\n\n")
    w("// The same stream with the panel's LDS loads and stores inside it (see gen_ls in the generator).  lb: LDS byte address\n")
    w("// of this lane's element of the panel's first column; sb / ss: where this lane writes column 0 and by how many bytes\n")
    w("// that address advances per column.\n")
    w("template <int C0>\n")
    w("__device__ __forceinline__ void potrf_panel16_chain_ls(unsigned lb, unsigned sb, unsigned ss, int diag_lane4) {\n")
    w("    float " + ", ".join("a%d" % i for i in range(16)) + ", d0, d1, t, s;\n")
    w("    asm volatile(\n")
    for ln in gen_ls():
        w('        "%s\\n\\t"\n' % ln)
    w("        : " + ", ".join('[a%d] "=&v"(a%d)' % (i, i) for i in range(16)) + ",\n")
    w('          [d0] "=&v"(d0), [d1] "=&v"(d1), [t] "=&v"(t), [s] "=&s"(s), [sb] "+v"(sb)\n')
    w('        : [addr] "v"(diag_lane4), [c0] "n"(C0), [lb] "v"(lb), [ss] "v"(ss), [p4] "n"(4 * PLD)\n')
    w('        : "memory");\n')
    w("}\n\n")
    w("// gen_ls3(): the same panel with the LDS latency off the pivot chain (two SGPR-path columns, counted waits);\n")
    w("// _ls3w: the same with wider fill slots behind the chain's links.\n")
    for name, slots in (("ls3", (1, 2, 1)), ("ls3w", (2, 3, 2))):
        w("template <int C0>\n")
        w("__device__ __forceinline__ void potrf_panel16_chain_%s(unsigned lb, unsigned sb, unsigned ss, int diag_lane4) {\n" % name)
        w("    float " + ", ".join("a%d" % i for i in range(16)) + ", " + ", ".join("d%d" % i for i in range(13)) + ", t, s, s2;\n")
        w("    asm volatile(\n")
        for ln in gen_ls3(slots):
            w('        "%s\\n\\t"\n' % ln)
        w("        : " + ", ".join('[a%d] "=&v"(a%d)' % (i, i) for i in range(16)) + ",\n")
        w("          " + ", ".join('[d%d] "=&v"(d%d)' % (i, i) for i in range(13)) + ",\n")
        w('          [t] "=&v"(t), [s] "=&s"(s), [s2] "=&s"(s2), [sb] "+v"(sb)\n')
        w('        : [addr] "v"(diag_lane4), [c0] "n"(C0), [lb] "v"(lb), [ss] "v"(ss), [p4] "n"(4 * PLD)\n')
        w('        : "memory");\n')
        w("}\n\n")

    if "--split" in sys.argv:   # round 5 experiment, not in the product (scripts/experiments/potrf64_lds_split.inc.txt)
        w("// Round 5: the panel split over two wavefronts (gen_split_diag / gen_split_x in the generator).  lbd: LDS byte address of\n")
        w("// element (C0 + lane % 16, C0) of the tile; exw: byte address of this lane's word of the exchange buffer's first column\n")
        w("// (Ex + 4 * lane; 256 bytes per column; word 16 of every column ZERO on entry: the flags).\n")
        w("template <int C0>\n")
        w("__device__ __forceinline__ void potrf_split_diag(unsigned lbd, unsigned exw) {\n")
        w("    float " + ", ".join("l%d" % i for i in range(16)) + ", t;\n")
        w("    asm volatile(\n")
        for ln in gen_split_diag():
            w('        "%s\\n\\t"\n' % ln)
        w("        : " + ", ".join('[l%d] "=&v"(l%d)' % (i, i) for i in range(16)) + ', [t] "=&v"(t)\n')
        w('        : [lbd] "v"(lbd), [exw] "v"(exw), [p4] "n"(4 * PLD)\n')
        w('        : "memory");\n')
        w("}\n\n")
        w("// x[j]: this lane's element of panel column C0 + j; exr = Ex + 4 * (lane % 16), ext = Ex + 64 (bytes); sb: where this lane writes\n")
        w("// column 0; potrf_split_x: column k goes to sb + k * 4 * PLD; potrf_split_x_strided: sb advances by the per-lane stride ss (bytes).\n")
        for name, imm in (("potrf_split_x", True), ("potrf_split_x_strided", False)):
            w("template <int C0>\n")
            w("__device__ __forceinline__ void %s(float (&x)[16], unsigned exr, unsigned ext, unsigned sb%s) {\n" % (name, "" if imm else ", unsigned ss"))
            w("    float " + ", ".join("t%d, l%d" % (i, i) for i in range(SPLIT_X_RING)) + ";\n")
            w("    asm volatile(\n")
            for ln in gen_split_x(imm):
                w('        "%s\\n\\t"\n' % ln)
            w("        : " + ", ".join('[x%d] "+v"(x[%d])' % (i, i) for i in range(16)) + ",\n")
            w("          " + ", ".join('[t%d] "=&v"(t%d), [l%d] "=&v"(l%d)' % (i, i, i, i) for i in range(SPLIT_X_RING)) + (', [sb] "+v"(sb)\n' if not imm else "\n"))
            w('        : [exr] "v"(exr), [ext] "v"(ext), ' + ('[sb] "v"(sb), [p4] "n"(4 * PLD)\n' if imm else '[ss] "v"(ss)\n'))
            w('        : "memory", "vcc");\n')
            w("}\n\n")


if __name__ == "__main__":
    main()
