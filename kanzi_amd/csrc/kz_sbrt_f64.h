// kz_sbrt_f64.h -- SBRT inverse (RANK / MTF / TIMESTAMP; K/transform/SBRT.java:154-214): the round-6 row form of k_sbrt_inverse
// (kz_sbrt.hip, which includes this file) for rows with many ranks >= 64 or many non-zero ranks, blocks up to 8 MiB.
//
// The list BY POSITION as in the other row forms, but every entry is ONE 64-bit key
//     hi = 0x40000000 | q                      lo = (p << 9) | (touched << 8) | (255 - symbol)
// read as an IEEE double: bit 62 is set, the exponent stays in 0x400..0x40F, so every key is a positive normal number and the order
// of the doubles is the order of the 64-bit integers.  That order IS the list order of the reference: q descending; among equal q
// the entry moved last first (the bubble loop of SBRT.java:203 passes every q <= qc, so a moved entry lands above its equals: larger
// p); entries never touched (q = p = 0) by ascending symbol, below the entry touched at i = 0 (the `touched` bit).  Keys are distinct.
// A step with rank r takes the entry at position r, gives it the key x of (qc, i) -- larger than its old key and than every key
// below -- and re-inserts it.  On a list sorted by key the whole move of SBRT.java:203-209 is, for every position j <= r,
//     new[j] = max(min(x, old[j-1]), old[j])          (old[-1] = +inf)
// (above the landing place min gives x > ... no: old[j] wins; at it x; below it old[j-1]; the entry's old key at j = r is dropped),
// i.e. one v_min_f64 + one v_max_f64 per register, the payload travelling inside the key: no threshold compare, no select pair, no
// second register to shift.  Positions above r are kept by the EXEC mask of a v_cmpx on the position number.
//
// Layout: position j sits in register pair (j & 3), lane (j >> 2) ("interleaved": old[j-1] is the neighbouring register of the same
// lane, one DPP wave_shr for register 0), so every rank runs the same straight-line code whatever its depth -- 27 instructions for
// RANK (the 32-bit forms of rounds 2-5: 22 for ranks below 64 in dense rows, 47 in the interleaved form, 40-90 in the row walk).
//   qc:  RANK (i + p) >> 1 = (lo + (i << 9)) >> 10  (one v_add3 + one v_alignbit that also ORs bit 30 in);  MTF i;  TIMESTAMP p = lo >> 9.
// p < 2^23 must hold for lo: blocks above 2^23 bytes keep the round-2 kernel (24-bit timestamps in a 32-bit word), blocks of 2^24 - 256
// and more the LDS form.
#define KZF_DPP " wave_shr:1 row_mask:0xf bank_mask:0xf\n\t"
// x.hi (v75) from the accessed entry's lo (s54); v55 = ((2 i + 1) << 8) of this step, v52 = -256, v60 scratch
#define KZF_X_RANK "v_add3_u32 v60, s54, v55, v52\n\t" "v_alignbit_b32 v75, %[c100], v60, 10\n\t"
// the same where i + p may reach 2^23 (the rows past 2^22 of a block: lo + (i << 9) no longer fits 32 bits): (lo >> 8) + 2 i, >> 2
#define KZF_X_RANK_HI "v_lshrrev_b32_e64 v60, 8, s54\n\t" "v_lshrrev_b32 v61, 8, v55\n\t" "v_add3_u32 v60, v60, v61, -1\n\t" "v_alignbit_b32 v75, 1, v60, 2\n\t"
#define KZF_X_MTF  "v_alignbit_b32 v75, %[c80], v55, 9\n\t"         /* i = ((2 i + 1) << 8) >> 9 (no counter of its own: x is computed under the step's EXEC mask in the by-position rows) */
#define KZF_X_TS   "v_mov_b32 v60, s54\n\t" "v_alignbit_b32 v75, %[c80], v60, 9\n\t"
// one rank at row position J (a constant); RC holds its rank, RN receives the next one
// (RC / LC / IC: the rank, its lane r >> 2 and its register offset 2 (r & 3), read from per-row vectors one step ahead)
#define KZF_STEP(J, JN, RC, RN, LC, LN, IC, IN, X)                 \
    "s_set_gpr_idx_on " IC ", gpr_idx(SRC0)\n\t"                     \
    "v_mov_b32 v54, v64\n\t"                     /* lo of register pair r & 3 */ \
    "s_set_gpr_idx_off\n\t"                                          \
    "v_mov_b32_dpp v72, v70" KZF_DPP                                 \
    "v_mov_b32_dpp v73, v71" KZF_DPP                                 \
    "v_readlane_b32 s54, v54, " LC "\n\t"                            \
    "v_readlane_b32 " RN ", %[cur], " #JN "\n\t"                     \
    "v_readlane_b32 " LN ", v62, " #JN "\n\t"                        \
    "v_readlane_b32 " IN ", v63, " #JN "\n\t"                        \
    "v_add_u32 v55, %[c512], v55\n\t"                                \
    X                                                                \
    "v_bfi_b32 v74, %[ff], s54, v55\n\t"                             \
    "v_writelane_b32 %[outv], s54, " #J "\n\t"                       \
    "v_min_f64 v[82:83], v[74:75], v[68:69]\n\t"                     \
    "v_min_f64 v[80:81], v[74:75], v[66:67]\n\t"                     \
    "v_min_f64 v[78:79], v[74:75], v[64:65]\n\t"                     \
    "v_min_f64 v[76:77], v[74:75], v[72:73]\n\t"                     \
    "v_cmpx_ge_u32 vcc, " RC ", v56\n\t"                             \
    "v_max_f64 v[64:65], v[76:77], v[64:65]\n\t"                     \
    "v_cmpx_ge_u32 vcc, " RC ", v57\n\t"                             \
    "v_max_f64 v[66:67], v[78:79], v[66:67]\n\t"                     \
    "v_cmpx_ge_u32 vcc, " RC ", v58\n\t"                             \
    "v_max_f64 v[68:69], v[80:81], v[68:69]\n\t"                     \
    "v_cmpx_ge_u32 vcc, " RC ", v59\n\t"                             \
    "v_max_f64 v[70:71], v[82:83], v[70:71]\n\t"                     \
    "s_mov_b64 exec, -1\n\t"
#define KZF_STEP2(A, B, C, X) KZF_STEP(A, B, "s42", "s43", "s44", "s46", "s45", "s47", X) KZF_STEP(B, C, "s43", "s42", "s46", "s44", "s47", "s45", X)
#define KZF_STEP8(A, B, C, D, E, F, G, H, I, X) KZF_STEP2(A, B, C, X) KZF_STEP2(C, D, E, X) KZF_STEP2(E, F, G, X) KZF_STEP2(G, H, I, X)
// one row of 64 ranks, every rank an ordinary step (a rank 0 re-keys the front entry in place)
#define KZF_ROW(X) asm volatile(                                                                 \
    "s_mov_b32 s41, m0\n\t"                                                                        \
    "v_mov_b32 v64, %[a0]\n\tv_mov_b32 v65, %[b0]\n\tv_mov_b32 v66, %[a1]\n\tv_mov_b32 v67, %[b1]\n\t" \
    "v_mov_b32 v68, %[a2]\n\tv_mov_b32 v69, %[b2]\n\tv_mov_b32 v70, %[a3]\n\tv_mov_b32 v71, %[b3]\n\t" \
    "v_mov_b32 v56, %[lane4]\n\tv_add_u32 v57, 1, %[lane4]\n\tv_add_u32 v58, 2, %[lane4]\n\tv_add_u32 v59, 3, %[lane4]\n\t" \
    "v_mov_b32 v72, 0\n\tv_mov_b32 v73, %[inf]\n\t"                                                \
    "v_mov_b32 v55, %[c2]\n\tv_mov_b32 v75, %[h0]\n\tv_mov_b32 v52, %[m256]\n\t"                   \
    "v_lshrrev_b32 v62, 2, %[cur]\n\tv_lshlrev_b32 v63, 1, %[cur]\n\tv_and_b32 v63, 6, v63\n\t"  /* lane and register offset of every rank of the row */ \
    "v_readlane_b32 s42, %[cur], 0\n\tv_readlane_b32 s44, v62, 0\n\tv_readlane_b32 s45, v63, 0\n\t" \
    "s_nop 3\n\t"                                    /* VALU-written SGPR -> lane select of v_readlane: 4 wait states */ \
    KZF_STEP8(0, 1, 2, 3, 4, 5, 6, 7, 8, X) KZF_STEP8(8, 9, 10, 11, 12, 13, 14, 15, 16, X)          \
    KZF_STEP8(16, 17, 18, 19, 20, 21, 22, 23, 24, X) KZF_STEP8(24, 25, 26, 27, 28, 29, 30, 31, 32, X) \
    KZF_STEP8(32, 33, 34, 35, 36, 37, 38, 39, 40, X) KZF_STEP8(40, 41, 42, 43, 44, 45, 46, 47, 48, X) \
    KZF_STEP8(48, 49, 50, 51, 52, 53, 54, 55, 56, X) KZF_STEP8(56, 57, 58, 59, 60, 61, 62, 63, 0, X)  \
    "v_mov_b32 %[a0], v64\n\tv_mov_b32 %[b0], v65\n\tv_mov_b32 %[a1], v66\n\tv_mov_b32 %[b1], v67\n\t" \
    "v_mov_b32 %[a2], v68\n\tv_mov_b32 %[b2], v69\n\tv_mov_b32 %[a3], v70\n\tv_mov_b32 %[b3], v71\n\t" \
    "s_mov_b32 m0, s41\n\t"                                                                        \
    : [a0]"+v"(A0), [b0]"+v"(B0), [a1]"+v"(A1), [b1]"+v"(B1), [a2]"+v"(A2), [b2]"+v"(B2), [a3]"+v"(A3), [b3]"+v"(B3), [outv]"+v"(outv) \
    : [cur]"v"(cur), [lane4]"v"(lane4), [ff]"v"(ff), [m256]"v"(m256), [inf]"v"(infHi), [c2]"v"(c2), [h0]"v"(h0), [c100]"s"(c100), [c80]"s"(c80), [c512]"s"(c512) \
    : "vcc", "scc", "v52", "v54", "v55", "v56", "v57", "v58", "v59", "v60", "v61", "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71",  \
      "v62", "v63", "v72", "v73", "v74", "v75", "v76", "v77", "v78", "v79", "v80", "v81", "v82", "v83", "s41", "s42", "s43", "s44", "s45", "s46", "s47", "s54");


// ---- rows without a rank >= 64, by-position layout (register pair k = positions 64 k .. 64 k + 63, one per lane) ----
// Only pair 0 moves: 13 instructions per rank (the dense rows of the 32-bit forms: 22; their row walk: ~40 per non-zero rank).
// (the chain of a step is: the entry read with v_readlane -> x -> v_min -> v_max; the shifted copies, the counter and the EXEC
// mask are issued while the read is on its way to the scalar registers; x and the minimum are only needed in the lanes <= r)
#define KZD_STEP(J, JN, RC, RN, X)                                 \
    "v_readlane_b32 s54, v64, " RC "\n\t"                            \
    "v_mov_b32_dpp v72, v64" KZF_DPP                                 \
    "v_mov_b32_dpp v73, v65" KZF_DPP                                 \
    "v_add_u32 v55, %[c512], v55\n\t"                                \
    "v_readlane_b32 " RN ", %[cur], " #JN "\n\t"                     \
    "v_cmpx_ge_u32 vcc, " RC ", v56\n\t"                             \
    X                                                                \
    "v_bfi_b32 v74, %[ff], s54, v55\n\t"                             \
    "v_min_f64 v[76:77], v[74:75], v[72:73]\n\t"                     \
    "v_max_f64 v[64:65], v[76:77], v[64:65]\n\t"                     \
    "s_mov_b64 exec, -1\n\t"                                         \
    "v_writelane_b32 %[outv], s54, " #J "\n\t"
#define KZD_STEP2(A, B, C, X) KZD_STEP(A, B, "s42", "s43", X) KZD_STEP(B, C, "s43", "s42", X)
#define KZD_STEP8(A, B, C, D, E, F, G, H, I, X) KZD_STEP2(A, B, C, X) KZD_STEP2(C, D, E, X) KZD_STEP2(E, F, G, X) KZD_STEP2(G, H, I, X)
#define KZD_ROW(X) asm volatile(                                                                 \
    "v_mov_b32 v64, %[a0]\n\tv_mov_b32 v65, %[b0]\n\t"                                             \
    "v_mov_b32 v56, %[lanev]\n\t"                                                                  \
    "v_mov_b32 v72, 0\n\tv_mov_b32 v73, %[inf]\n\t"                                                \
    "v_mov_b32 v55, %[c2]\n\tv_mov_b32 v75, %[h0]\n\tv_mov_b32 v52, %[m256]\n\t"                   \
    "v_readlane_b32 s42, %[cur], 0\n\t"                                                            \
    "s_nop 3\n\t"                                    /* VALU-written SGPR -> lane select of v_readlane: 4 wait states */ \
    KZD_STEP8(0, 1, 2, 3, 4, 5, 6, 7, 8, X) KZD_STEP8(8, 9, 10, 11, 12, 13, 14, 15, 16, X)          \
    KZD_STEP8(16, 17, 18, 19, 20, 21, 22, 23, 24, X) KZD_STEP8(24, 25, 26, 27, 28, 29, 30, 31, 32, X) \
    KZD_STEP8(32, 33, 34, 35, 36, 37, 38, 39, 40, X) KZD_STEP8(40, 41, 42, 43, 44, 45, 46, 47, 48, X) \
    KZD_STEP8(48, 49, 50, 51, 52, 53, 54, 55, 56, X) KZD_STEP8(56, 57, 58, 59, 60, 61, 62, 63, 0, X)  \
    "v_mov_b32 %[a0], v64\n\tv_mov_b32 %[b0], v65\n\t"                                             \
    : [a0]"+v"(A0), [b0]"+v"(B0), [outv]"+v"(outv)                                                 \
    : [cur]"v"(cur), [lanev]"v"(lanev), [ff]"v"(ff), [m256]"v"(m256), [inf]"v"(infHi), [c2]"v"(c2), [h0]"v"(h0), [c100]"s"(c100), [c80]"s"(c80), [c512]"s"(c512) \
    : "vcc", "scc", "v52", "v55", "v56", "v60", "v61", "v64", "v65", "v72", "v73", "v74", "v75", "v76", "v77", "s42", "s43", "s54");

// The same rows with a few ranks >= 64 among them: every step tests its rank (15 instructions), a rank >= 64 leaves through a stub
// to ONE row-spanning step (all four pairs: the entry is fetched with a register-indexed v_mov, rows 1..3 take the last entry of
// the row above into lane 0 of their shifted neighbours) and comes back with s_setpc: ~50 instructions + two taken branches.
#define KZD_STEPC(J, JN, RC, RN, X)                                \
    "v_readlane_b32 " RN ", %[cur], " #JN "\n\t"                     \
    "s_cmp_ge_u32 " RC ", 64\n\t"                                    \
    "s_cbranch_scc1 L_stub" #J "_%=\n\t"                             \
    "v_readlane_b32 s54, v64, " RC "\n\t"                            \
    "v_mov_b32_dpp v72, v64" KZF_DPP                                 \
    "v_mov_b32_dpp v73, v65" KZF_DPP                                 \
    "v_add_u32 v55, %[c512], v55\n\t"                                \
    "v_cmpx_ge_u32 vcc, " RC ", v56\n\t"                             \
    X                                                                \
    "v_bfi_b32 v74, %[ff], s54, v55\n\t"                             \
    "v_min_f64 v[76:77], v[74:75], v[72:73]\n\t"                     \
    "v_max_f64 v[64:65], v[76:77], v[64:65]\n\t"                     \
    "s_mov_b64 exec, -1\n\t"                                         \
    "v_writelane_b32 %[outv], s54, " #J "\n\t"                       \
  "L_after" #J "_%=:\n\t"
#define KZD_STUB(J, RC)                                            \
  "L_stub" #J "_%=:\n\t"                                             \
    "s_mov_b32 s40, " #J "\n\t"                                      \
    "s_mov_b32 s46, " RC "\n\t"                                      \
    "s_getpc_b64 s[62:63]\n\t"                                       \
  "L_pc" #J "_%=:\n\t"                                               \
    "s_sub_u32 s62, s62, L_pc" #J "_%=-L_after" #J "_%=\n\t"           \
    "s_subb_u32 s63, s63, 0\n\t"                                     \
    "s_branch L_deep_%=\n\t"
#define KZD_STEPC2(A, B, C, X) KZD_STEPC(A, B, "s42", "s43", X) KZD_STEPC(B, C, "s43", "s42", X)
#define KZD_STEPC8(A, B, C, D, E, F, G, H, I, X) KZD_STEPC2(A, B, C, X) KZD_STEPC2(C, D, E, X) KZD_STEPC2(E, F, G, X) KZD_STEPC2(G, H, I, X)
#define KZD_STUB2(A, B) KZD_STUB(A, "s42") KZD_STUB(B, "s43")
#define KZD_STUB8(A, B, C, D, E, F, G, H) KZD_STUB2(A, B) KZD_STUB2(C, D) KZD_STUB2(E, F) KZD_STUB2(G, H)
// the row-spanning step: s46 = rank (>= 64), s40 = row position; v[84:89] = shifted neighbours of pairs 1..3, v[78:83] their minima
#define KZD_CARRY(NLO, NHI, SLO, SHI, PLO, PHI)                    \
    "v_readlane_b32 s55, " PLO ", 63\n\t"                            \
    "v_readlane_b32 s56, " PHI ", 63\n\t"                            \
    "v_mov_b32_dpp " NLO ", " SLO KZF_DPP                            \
    "v_mov_b32_dpp " NHI ", " SHI KZF_DPP                            \
    "v_writelane_b32 " NLO ", s55, 0\n\t"                            \
    "v_writelane_b32 " NHI ", s56, 0\n\t"
#define KZD_DEEP(X)                                                \
  "L_deep_%=:\n\t"                                                   \
    "s_lshr_b32 s45, s46, 5\n\t"                                     \
    "s_and_b32 s45, s45, 6\n\t"                  /* 2 x (rank >> 6): the pair's first register */ \
    "s_and_b32 s44, s46, 63\n\t"                                     \
    "s_set_gpr_idx_on s45, gpr_idx(SRC0)\n\t"                        \
    "v_mov_b32 v54, v64\n\t"                                         \
    "s_set_gpr_idx_off\n\t"                                          \
    "v_add_u32 v55, %[c512], v55\n\t"                                \
    "v_readlane_b32 s54, v54, s44\n\t"                               \
    KZD_CARRY("v84", "v85", "v66", "v67", "v64", "v65")             \
    KZD_CARRY("v86", "v87", "v68", "v69", "v66", "v67")             \
    KZD_CARRY("v88", "v89", "v70", "v71", "v68", "v69")             \
    "v_mov_b32_dpp v72, v64" KZF_DPP                                 \
    "v_mov_b32_dpp v73, v65" KZF_DPP                                 \
    X                                                                \
    "v_bfi_b32 v74, %[ff], s54, v55\n\t"                             \
    "s_mov_b32 m0, s40\n\t"                                          \
    "v_min_f64 v[76:77], v[74:75], v[72:73]\n\t"                     \
    "v_min_f64 v[78:79], v[74:75], v[84:85]\n\t"                   \
    "v_min_f64 v[80:81], v[74:75], v[86:87]\n\t"                   \
    "v_min_f64 v[82:83], v[74:75], v[88:89]\n\t"                   \
    "v_writelane_b32 %[outv], s54, m0\n\t"                           \
    "v_max_f64 v[64:65], v[76:77], v[64:65]\n\t"  /* row 0 lies above every rank >= 64: all lanes */ \
    "v_cmpx_ge_u32 vcc, s46, v57\n\t"                                \
    "v_max_f64 v[66:67], v[78:79], v[66:67]\n\t"                     \
    "v_cmpx_ge_u32 vcc, s46, v58\n\t"                                \
    "v_max_f64 v[68:69], v[80:81], v[68:69]\n\t"                     \
    "v_cmpx_ge_u32 vcc, s46, v59\n\t"                                \
    "v_max_f64 v[70:71], v[82:83], v[70:71]\n\t"                     \
    "s_mov_b64 exec, -1\n\t"                                         \
    "s_setpc_b64 s[62:63]\n\t"
#define KZD_ROWC(X) asm volatile(                                                                \
    "s_mov_b32 s41, m0\n\t"                                                                        \
    "v_mov_b32 v64, %[a0]\n\tv_mov_b32 v65, %[b0]\n\tv_mov_b32 v66, %[a1]\n\tv_mov_b32 v67, %[b1]\n\t" \
    "v_mov_b32 v68, %[a2]\n\tv_mov_b32 v69, %[b2]\n\tv_mov_b32 v70, %[a3]\n\tv_mov_b32 v71, %[b3]\n\t" \
    "v_mov_b32 v56, %[lanev]\n\tv_add_u32 v57, 64, %[lanev]\n\tv_add_u32 v58, %[c128], %[lanev]\n\tv_add_u32 v59, %[c192], %[lanev]\n\t" \
    "v_mov_b32 v72, 0\n\tv_mov_b32 v73, %[inf]\n\t"                                                \
    "v_mov_b32 v55, %[c2]\n\tv_mov_b32 v75, %[h0]\n\tv_mov_b32 v52, %[m256]\n\t"                   \
    "v_readlane_b32 s42, %[cur], 0\n\t"                                                            \
    "s_nop 3\n\t"                                                                                  \
    KZD_STEPC8(0, 1, 2, 3, 4, 5, 6, 7, 8, X) KZD_STEPC8(8, 9, 10, 11, 12, 13, 14, 15, 16, X)        \
    KZD_STEPC8(16, 17, 18, 19, 20, 21, 22, 23, 24, X) KZD_STEPC8(24, 25, 26, 27, 28, 29, 30, 31, 32, X) \
    KZD_STEPC8(32, 33, 34, 35, 36, 37, 38, 39, 40, X) KZD_STEPC8(40, 41, 42, 43, 44, 45, 46, 47, 48, X) \
    KZD_STEPC8(48, 49, 50, 51, 52, 53, 54, 55, 56, X) KZD_STEPC8(56, 57, 58, 59, 60, 61, 62, 63, 0, X)  \
    "s_branch L_done_%=\n\t"                                                                       \
    KZD_STUB8(0, 1, 2, 3, 4, 5, 6, 7) KZD_STUB8(8, 9, 10, 11, 12, 13, 14, 15) KZD_STUB8(16, 17, 18, 19, 20, 21, 22, 23) \
    KZD_STUB8(24, 25, 26, 27, 28, 29, 30, 31) KZD_STUB8(32, 33, 34, 35, 36, 37, 38, 39) KZD_STUB8(40, 41, 42, 43, 44, 45, 46, 47) \
    KZD_STUB8(48, 49, 50, 51, 52, 53, 54, 55) KZD_STUB8(56, 57, 58, 59, 60, 61, 62, 63)              \
    KZD_DEEP(X)                                                                                    \
  "L_done_%=:\n\t"                                                                                 \
    "v_mov_b32 %[a0], v64\n\tv_mov_b32 %[b0], v65\n\tv_mov_b32 %[a1], v66\n\tv_mov_b32 %[b1], v67\n\t" \
    "v_mov_b32 %[a2], v68\n\tv_mov_b32 %[b2], v69\n\tv_mov_b32 %[a3], v70\n\tv_mov_b32 %[b3], v71\n\t" \
    "s_mov_b32 m0, s41\n\t"                                                                        \
    : [a0]"+v"(A0), [b0]"+v"(B0), [a1]"+v"(A1), [b1]"+v"(B1), [a2]"+v"(A2), [b2]"+v"(B2), [a3]"+v"(A3), [b3]"+v"(B3), [outv]"+v"(outv) \
    : [cur]"v"(cur), [lanev]"v"(lanev), [ff]"v"(ff), [m256]"v"(m256), [inf]"v"(infHi), [c2]"v"(c2), [h0]"v"(h0), [c100]"s"(c100), [c80]"s"(c80), [c512]"s"(c512), [c128]"s"(c128), [c192]"s"(c192) \
    : "vcc", "scc", "v52", "v54", "v55", "v56", "v57", "v58", "v59", "v60", "v61", "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71",  \
      "v72", "v73", "v74", "v75", "v76", "v77", "v78", "v79", "v80", "v81", "v82", "v83", "v84", "v85", "v86", "v87", "v88", "v89", \
      "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s54", "s55", "s56", "s62", "s63");

// ---- the two representations of an entry (per entry: the same in both layouts) ----
//   32-bit forms: Q = (q << 8) | symbol, P = p                    64-bit key: lo = (p << 9) | (touched << 8) | (255 - symbol), hi = 0x40000000 | q
// touched: the entry has been accessed.  Q and P show it (q or p non-zero) except for the entry accessed at i = 0 and not since
// (q = p = 0 like the untouched ones: its place above them is its position in the 32-bit forms): that entry holds symbol s0, the
// first symbol decoded (s0 < 0: nothing decoded yet).
__device__ __forceinline__ void kzf_pack(uint32_t& Q, uint32_t& P, int s0) {
  const uint32_t sym = Q & 0xFFu, q = Q >> 8;
  const uint32_t touched = ((q | P) != 0u || (int)sym == s0) ? 0x100u : 0u;
  Q = (P << 9) | touched | (255u - sym);
  P = 0x40000000u | q;
}
__device__ __forceinline__ void kzf_unpack(uint32_t& A, uint32_t& B) {
  const uint32_t sym = 255u - (A & 0xFFu), p = A >> 9;
  A = ((B & 0x00FFFFFFu) << 8) | sym;
  B = p;
}
