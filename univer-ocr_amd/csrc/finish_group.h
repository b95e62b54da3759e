// Deferred finish kernels of the weight-gradient producers (finish_group.hip).
//
// Every direct weight-gradient kernel ends in block partials [block][accumulator] (float32) and a small "finish" launch
// that adds the columns in float64 and writes dw / db.  Inside the page step even a trivial launch costs the lane that
// issues it 8-10 us (profiles/r03_*: upconv_weff_kernel, one block, 9.9 us), and the Paragraph and Line nets each have
// five of these per step.  Between uocr_wgrad_defer_begin and _flush the producers take their partial buffer from a
// region of its own (it has to survive the kernels that follow) and only RECORD their finish; the flush runs all of
// them as ONE launch: block -> (recorded finish, 8 columns), 32 row segments per column as in pair_strip_finish, then
// the column's destination by the producer's layout rule.  Same sums in the same order as the separate kernels.
#pragma once
#include "uocr_common.h"

enum : int {
    FIN_COLS = 0,      // column a -> dw[a] (a < ndw), db[a - ndw]                                  p: ndw
    FIN_TAPROWS = 1,   // [tap row][NP]: idx < NW -> dw[ky * NW + idx]; tap row 0, idx < NW + NB -> db   p: NP, NW, NB
    FIN_FAST = 2,      // conv_wgrad_fast: groups (kyg, ocg) of NP columns                          p: NP, NW, KW, CIN, COUT, KYR, COB
    FIN_PAIR = 3,      // the Monochrome pair block: dw1 (9 x 16), db1, dw2, db2 at fixed columns
};

struct FinishDesc {
    int kind;
    const float* partial;
    int nblocks;            // rows
    int ncols;              // columns, all groups
    int group_cols;         // columns per group (= ncols when the matrix is one piece)
    size_t group_stride;    // floats between the groups' matrices
    size_t row_stride;      // floats between rows
    float *dw, *db, *dw2, *db2;
    int use_bias, use_bias2, accumulate;
    float unscale;
    int p[7];
    int first_block;        // (filled in by the flush)
};

// partial buffer of `bytes` for a producer: from the deferred region when a group is open and it fits (then the finish
// MUST be offered to uocr_finish_defer), else the ctx workspace (nullptr + error code when that is too small)
float* uocr_partial_buffer(uocr_ctx* ctx, size_t bytes, int* rc);
// true: recorded -- the caller returns without launching its finish kernel
bool uocr_finish_defer(uocr_ctx* ctx, const FinishDesc& d);
void uocr_finish_defer_free(uocr_ctx* ctx);
