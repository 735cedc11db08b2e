// jv_xb.h — argument blocks of the batched exact scorer's kernels (jv_kernels_xb.hip), shared with jv_abi.cpp.
#pragma once
#include <stdint.h>

// one launch of the MFMA tile kernel: `rows` candidates x B queries
struct JvXbTileArgs {
    const uint16_t* vb;     // bf16 mirror of the vectors [n][kp]
    const float* vnorm2;    // [n] |v|^2 (fp32 sum of squares of the fp32 row)
    int32_t kp;             // k padded to a multiple of 64
    int32_t n;              // rows of the mirror: a list entry outside [0, n) is skipped (never a survivor, lower bound -inf)
    const int32_t* ords;    // the shared candidate list (ordinals) or nullptr = ordinal i is candidate i
    int32_t C;              // its length
    int32_t cstride;        // candidate j of THIS launch is list entry min(j * cstride, C - 1)  (1 = all; > 1 = the strided sample)
    int32_t rows;           // candidates of this launch
    const uint16_t* qb;     // bf16 queries [panels * 128][kp], zero rows behind B
    const float* qnorm2;    // [B]
    int32_t B, panels;
    int32_t qbase;          // query-stationary kernel: first query of this launch's round of <= 256
    unsigned long long* stamps;  // diagnostics (JV_XB_STAMPS): 8 cycle accumulators of wave 1 of every workgroup, or nullptr
    int32_t dbg;            // diagnostics (JV_XB_DBG): bit 0 skips the multiply loop, bit 1 the epilogue — timing only, wrong answers
    int32_t sim;
    float kappa;            // relative half-width of the bf16 product's error, in units of |q||c|
    float* sample;          // mode 0: [B][sample_ld] lower bounds
    int32_t sample_ld;
    const float* thr;       // mode 1: [B] per-query bar (k-th largest lower bound of the sample)
    int32_t* surv_cnt;      // mode 1: [B] survivors appended (may exceed surv_cap: the query is then re-scored against the whole list)
    int32_t* surv;          // mode 1: [B][surv_cap] list positions
    int32_t surv_cap;
};

struct JvXbRescoreArgs {
    const float* queries;   // [B][d] fp32
    const int32_t* ords;    // candidate list or nullptr (identity)
    int32_t C;
    const int32_t* surv_cnt;  // nullptr = no pre-filter ran: every query scans the whole list
    const int32_t* surv;
    int32_t surv_cap;
    int32_t force_all;
    int32_t topK;
    int32_t* out_nodes;     // [B][topK] ordinals (-1 = empty)
    int32_t* out_docs;      // [B][topK]
    float* out_scores;      // [B][topK]
    int32_t* out_count;     // [B]
    int64_t* out_info;      // optional device words: [0] += rows re-scored, [1] += queries that scanned the whole list after an overflow
};
