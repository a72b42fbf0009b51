// Diagnostics builds of libmpcmax (never part of the product library): per-phase time stamps of single kernels, read back by the
// probes in tools/ (bucket_stamp_probe.py, more_stamp_probe.py, bwd_stamp_probe.py, lut_accum_stamp_probe.py, inbin_probe.py).
// Each is switched on by its own -D flag through MPC_EXTRA_HIPCC_FLAGS; without the flag every macro below expands to nothing, so
// the kernels in the product translation units carry only the one-word markers (BK_STAMP(3); KS_STP(); ...).
// Stamps are taken by thread 0 behind an s_waitcnt(0), in 10 ns units of wall_clock64(), and land far inside the KNN forward's
// `fail` list (a region no product run reaches).
#pragma once

// ---- -DKNN_BK_STAMP: phases of k_knn_bucket (tools/bucket_stamp_probe.py) --------------------------------------------------
#ifdef KNN_BK_STAMP
#define BK_STAMP_DECL __shared__ unsigned s_stp[8];
#define BK_STAMP(k) do { __builtin_amdgcn_s_waitcnt(0); if (threadIdx.x == 0) s_stp[k] = (unsigned)wall_clock64(); } while (0)
#define BK_STAMP_WRITE(ls_, tid_) do { BK_STAMP(7); __syncthreads(); \
        if ((tid_) < 8) (ls_).fail[1 + 100000 + 8 * blockIdx.x + (tid_)] = (tid_) == 0 ? (int)s_stp[0] : (int)(s_stp[tid_] - s_stp[0]); } while (0)
#else
#define BK_STAMP_DECL
#define BK_STAMP(k) do { } while (0)
#define BK_STAMP_WRITE(ls_, tid_) do { } while (0)
#endif

// ---- -DKS_STAMP2: phases of a work item of the far pass of k_knn_tail (tools/more_stamp_probe.py) --------------------------
// (-DKS_STAMP0, round 6: the same stamps for the MAIN launch, one record per strip workgroup: tools/strip_stamp_probe.py)
#if defined(KS_STAMP2) || defined(KS_STAMP0)
#ifdef KS_STAMP0
#define KS_STP_ON (MODE == 0)
#else
#define KS_STP_ON (FARK)
#endif
#define KS_STP_DECL unsigned long long stp_[8]; int nstp_ = 0;
#define KS_STP() do { if (KS_STP_ON && nstp_ < 8) { __builtin_amdgcn_s_waitcnt(0); stp_[nstp_++] = wall_clock64(); } } while (0)
#define KS_STP_WRITE(ls_, tid_, mine_, total_) do { if (KS_STP_ON) { __syncthreads(); KS_STP(); const int nmk_ = __syncthreads_count((mine_) ? 1 : 0); \
        if ((tid_) == 0) { int *dst_ = (ls_).fail + 1 + 200000 + 12 * (int)blockIdx.x; dst_[0] = (int)(stp_[0] & 0x7fffffffull); \
            for (int k_ = 1; k_ < 8; ++k_) dst_[k_] = (int)(stp_[k_] - stp_[0]); dst_[8] = (total_); dst_[9] = nmk_; } } } while (0)
#else
#define KS_STP_DECL
#define KS_STP() do { } while (0)
#define KS_STP_WRITE(ls_, tid_, mine_, total_) do { } while (0)
#endif

// ---- -DKS_DEBUG_INBIN: statistics of the strip kernel's fast path in the (otherwise unused) normaliser plane (tools/inbin_probe.py)
#ifdef KS_DEBUG_INBIN
#define KS_INBIN_STAT(knn_state_, BQ_, q_, inbin_, nsl_, iwd_, norm_) do { (knn_state_)[2 * (BQ_) + (q_)] = (float)(inbin_) + 100.f * (float)(nsl_); } while (0)
#else
#define KS_INBIN_STAT(knn_state_, BQ_, q_, inbin_, nsl_, iwd_, norm_) do { if (iwd_) (knn_state_)[2 * (BQ_) + (q_)] = (norm_); } while (0)
#endif

// ---- -DKNN_BW_STAMP: lifetime / reach phase of every workgroup of k_knn_bwd_tile (tools/bwd_stamp_probe.py) ---------------
#ifdef KNN_BW_STAMP
#define KB_STAMP_PARAM , int *__restrict__ stamp
#define KB_STAMP_BEGIN const unsigned long long st0 = wall_clock64(); int st_slow = 0;
#define KB_STAMP_MID const unsigned long long st1 = wall_clock64();
#define KB_STAMP_SLOW st_slow = 1;
#define KB_STAMP_END(tid_, lblk_, RQ_, use_lds_, anytie_, total_) do { __syncthreads(); const bool anyslow_ = __syncthreads_or(st_slow) != 0; \
        if ((tid_) == 0) { const unsigned long long st2_ = wall_clock64(); stamp[4 * (lblk_) + 0] = (int)(st2_ - st0); stamp[4 * (lblk_) + 1] = (int)(st1 - st0); \
            stamp[4 * (lblk_) + 2] = (RQ_) | ((use_lds_) ? 0 : 256) | ((anytie_) ? 512 : 0) | (anyslow_ ? 1024 : 0); stamp[4 * (lblk_) + 3] = (total_); } } while (0)
#define KB_STAMP_ARG(ws_, L_) , (int *)((char *)(ws_) + (L_).off_knn_fail)
#else
#define KB_STAMP_PARAM
#define KB_STAMP_BEGIN
#define KB_STAMP_MID
#define KB_STAMP_SLOW
#define KB_STAMP_END(tid_, lblk_, RQ_, use_lds_, anytie_, total_) do { } while (0)
#define KB_STAMP_ARG(ws_, L_)
#endif

// ---- -DEV_LA_STAMP: phases of wavefront 0 of k_lut_accum (tools/lut_accum_stamp_probe.py) ----------------------------------
#ifdef EV_LA_STAMP
// (in LDS, not registers: eight live 64-bit values took the kernel from 47 to > 64 VGPRs -- one workgroup per CU)
#define LA_STAMP_DECL __shared__ unsigned s_stp[8]; if (threadIdx.x < 8) s_stp[threadIdx.x] = (unsigned)wall_clock64();
#define LA_STAMP(k) do { __builtin_amdgcn_s_waitcnt(0); if (threadIdx.x == 0) s_stp[k] = (unsigned)wall_clock64(); } while (0)
#define LA_STAMP_WRITE(tid_, dst_, n_) do { LA_STAMP(6); __syncthreads(); if ((tid_) == 0) { unsigned *d_ = reinterpret_cast<unsigned *>(dst_); \
        for (int k_ = 0; k_ < 7; ++k_) d_[k_] = s_stp[k_] - s_stp[0]; d_[7] = (unsigned)(n_); d_[8] = s_stp[0]; } } while (0)
#else
#define LA_STAMP_DECL
#define LA_STAMP(k) do { } while (0)
#define LA_STAMP_WRITE(tid_, dst_, n_) do { } while (0)
#endif

// ---- -DKT_TIMELINE: when the workgroups of k_knn_tail start, finish their lists and end (tools/tail_timeline_probe.py) -----------
// every workgroup writes 8 ints far inside the `fail` list: [0] start, [1] strip workgroups: end of the items / fallback: main list done,
// [2] fallback: wait over, [3] end (10 ns units of wall_clock64, low 31 bits), [4] items or entries of the main list this workgroup's
// wavefront 0 took, [5] late entries it took
#ifdef KT_TIMELINE
#define KT_T(k_) do { __builtin_amdgcn_s_waitcnt(0); if (threadIdx.x == 0) kt_[k_] = (int)(wall_clock64() & 0x7fffffffull); } while (0)
#define KT_DECL int kt_[6] = {0, 0, 0, 0, 0, 0};
#define KT_COUNT(k_, n_) do { kt_[k_] = (n_); } while (0)
#define KT_WRITE(ls_) do { if (threadIdx.x == 0) { int *d_ = (ls_).fail + 1 + 300000 + 8 * (int)blockIdx.x; for (int k_ = 0; k_ < 6; ++k_) d_[k_] = kt_[k_]; } } while (0)
#else
#define KT_T(k_) do { } while (0)
#define KT_DECL
#define KT_COUNT(k_, n_) do { } while (0)
#define KT_WRITE(ls_) do { } while (0)
#endif
