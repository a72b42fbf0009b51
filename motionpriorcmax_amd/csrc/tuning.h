// Compile-time constants of libmpcmax that a tuning sweep may override (tools/sweep_*.sh, tools/ab_*.sh set MPC_EXTRA_HIPCC_FLAGS=-DNAME=value;
// never set for a product build): every one with the measurement that chose its value.  Collected here (round 5) so that the
// translation units read as product code; the values are the ones the kernels were profiled with.
#pragma once

// ---- contrast.hip --------------------------------------------------------------------------------------------------
#ifndef CM_PF
#define CM_PF 4      // rows of the raw image in flight ahead of the row being processed (2: 35.5 us at C3, 4: 34.7, 8: 34.3)
#endif
#ifndef MPC_SM_PF
#define MPC_SM_PF 2
#endif

// ---- events.hip ----------------------------------------------------------------------------------------------------
#ifndef EV_PER_THREAD
#define EV_PER_THREAD 2   // measured at C3: 2 -> 98.6 us, 4 -> 106.9, 8 -> 107.5 (whole forward splat)
#endif
#ifndef EV_STAGE
#define EV_STAGE 1152      // records of one workgroup laid out in LDS before they are written (>= 2.25 per event)
#endif
#ifndef EV_LUT_THREADS
#define EV_LUT_THREADS 512
#endif
#ifndef EV_LUT_MINWAVES
#define EV_LUT_MINWAVES 6    // wavefronts per SIMD the register budget of k_lut_accum<false> is set for: 6 = three 512-thread workgroups per CU (what
                             // its 46 KB LDS strips are sized for); the kernel had drifted to 81 VGPRs -- one over the 80 of six wavefronts, i.e. TWO
                             // workgroups per CU (round 6, see profiles/HISTORY_r06.md)
#endif
#ifndef EV_LUT_THREADS_ORD
#define EV_LUT_THREADS_ORD 1024      // ordered variant: 72 KB of LDS per workgroup -> two per CU; 1024 threads keep the CU's waves
#endif
#ifndef EV_LUT_INFLIGHT_ORD
#define EV_LUT_INFLIGHT_ORD 2
#endif
#ifndef EV_LUT_INFLIGHT
#define EV_LUT_INFLIGHT 4   // records of a thread in flight together.  Round 3, stage call: 4: 43.8 us, 5: 42.4, 6: 43.6, 8: 52.3; round 6, inside the step and at
                            // three workgroups per CU (EV_LUT_MINWAVES): 4: 55.9 us at C3 / 44.7 at C4 batch 6, 5: 56.9 / 45.7, 6: 63.0 / 51.2
#endif

// ---- knn.hip -------------------------------------------------------------------------------------------------------
#ifndef KNN_BW_CH
#define KNN_BW_CH 4       // bwd_window_fast: cells of a window row whose LDS reads are issued together
#endif
#ifndef KNN_BW_PITCH
#define KNN_BW_PITCH 48   // row pitch (cells) of the staged arrays without the flow_to_next gradient; >= 16 + 2 * KNN_RQ_MAX
#endif
#ifndef KNN_BW_PITCH_NEXT
#define KNN_BW_PITCH_NEXT 32   // ... with the flow_to_next gradient (20 bytes per cell)
#endif
#ifndef KNN_BW_OCC
#define KNN_BW_OCC 7
#endif
#ifndef KNN_BW_OCC_IWD
#define KNN_BW_OCC_IWD 7      // 'iwd' without the flow_to_next gradient (12 bytes of LDS per cell as for 'mean'): k_knn_bwd_tile at C3 155.1 us at six per CU, 148.6 at seven
#endif
#ifndef KNN_BW_OCC_NEXT
#define KNN_BW_OCC_NEXT 6     // with the flow_to_next gradient (and for 'iwd'): 80 VGPRs, 23 KB of LDS per workgroup.  C4 batch 6, k_knn_bwd_tile:
                              // 8 per CU by registers (64, 31 spilled; the LDS -- 24 bytes per cell then -- allowed 5) 242 us; the K-th index
                              // out of the LDS (20 bytes per cell: 6 per CU) at 8 / 6 / 5: 218 / 206 / 215; + the conflict-free pitch where the
                              // region leaves room (k_knn_bwd_tile: RP) at 7 / 6 / 5: 209 / 202 / 212  (profiles/r06_ab_bwd_next.txt)
#endif
#ifndef KNN_FAR_QB
#define KNN_FAR_QB 256        // far queries tested against the tile per batch (one per thread); those that touch it: a list in LDS
                              // (512 / 1024 per batch: 244 / 273 us against 237 at a 48 px contraction band)
#endif
#ifndef KNN_FAR_BLOCKS
#define KNN_FAR_BLOCKS 4096     // (2048: 237 us against 218 at a 48 px contraction band)
#endif

// ---- knn_strip.hip -------------------------------------------------------------------------------------------------
#ifndef KS_MORE_MIN
#define KS_MORE_MIN 16               // a strip goes to the second launch for its unfinished queries if it has more than this many (or far queries)
#endif
#ifndef KS_MAIN_CHORD
#define KS_MAIN_CHORD 1            // main launch: region rows as wide as the widest chord (1) or square (0) that uses them
#endif
#ifndef KS_MAXCH_FAR
#define KS_MAXCH_FAR 48             // ... of the launch for the far queries (192 slots: a band along the left or right border -- every region row
                                   // as wide as the widest chord -- needs ~160; 256 slots at three workgroups per CU measured slower, see KS_MORE_OCC)
#endif
#ifndef KS_SB
#define KS_SB 3                     // staging: items per thread whose global loads are in flight together
#endif
#ifndef KS_FORWARD_MAX
#define KS_FORWARD_MAX 1024         // the main launch marked at most this many queries: the tail's one-wavefront search takes them from the marked list, no far pass
#endif
#ifndef KS_MORE_OCC
#define KS_MORE_OCC 4               // workgroups per CU of the second launch = its register budget (128; 13 registers spill).  The launch is bound
                                   // by the latency of its work items: C3, 40 px translation / 48 px contraction band: 4 per CU with 192 slots
                                   // 161 / 308 us, 3 per CU with 192 slots 179 / 351, with 256 slots 190 / 368, 2 per CU 258 / -
#endif

// ---- voxel.hip -----------------------------------------------------------------------------------------------------
#ifndef VOX_STRIP_KB
#define VOX_STRIP_KB 75       // LDS budget of a strip: two workgroups per CU, whose zero / accumulate / write phases overlap (150 KB, one per CU:
                              // 0.283 ms against 0.265 at the DSEC batch shape; 50 KB: the binning pass pays for the extra buckets)
#endif

// ---- common.h ------------------------------------------------------------------------------------------------------
#ifndef MPC_CT_H
#define MPC_CT_H 32
#endif

// ---- knn_device.h --------------------------------------------------------------------------------------------------
#ifndef KNN_BINS
#define KNN_BINS 32
#endif
#ifndef KNN_BATCH
#define KNN_BATCH 2   // candidate positions loaded ahead of use in the two hot scans (B=14: 944 -> 874 us; 4: 867)
#endif
#ifndef KNN_RCAP
#define KNN_RCAP 6                 // largest search radius (cells) of the strip kernel's main launch
#endif
#ifndef KNN_FAR_RINGS
#define KNN_FAR_RINGS 2            // a K-th distance beyond r_init + this many rings: the far backward's query (knn_is_far_dk)
#endif

