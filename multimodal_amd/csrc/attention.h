// Parameter block shared by the attention kernels (attention.hip, attention_fast.hip).
#pragma once
#include "common.h"

struct AttnP {
    const void* q; const void* k; const void* v; void* o; float* lse;
    const void* dO; void* dq; void* dk; void* dv; const float* delta;
    const float* cos_t; const float* sin_t;
    int B, H, S, d, rot, P;
    long ld, ld_o, ld_d;
    float scale;
    int vec_ok;
};

// bf16 fast forms (attention_fast.hip): pre-rotated q,k (rot == 0), 8-byte aligned rows, d % 4 == 0, d <= 192.
// Return MM_OK after launching, or a negative code; `attn_fast_ok` says whether the shape is covered.
bool attn_fast_ok(const AttnP& p);
int attn_fast_fwd(const AttnP& p, int mode, hipStream_t st);
int attn_fast_bwd(const AttnP& p, int mode, hipStream_t st);      // dq and dk/dv kernels (delta already computed)

// bf16 forms for head_dim = 512 (attention_wide.hip): pre-rotated q,k, 16-byte aligned rows; DMA-staged tiles, one wave per SIMD.
bool attn_wide_ok(const AttnP& p);
int attn_wide_fwd(const AttnP& p, int mode, hipStream_t st);
int attn_wide_bwd(const AttnP& p, int mode, hipStream_t st);      // dq and dk/dv kernels (delta already computed)

// bf16 forms with a whole head's streamed operands resident in LDS (attention_res.hip): head_dim <= 128 and S small enough
// (S = 315 at d = 100, S = 238 at d = 128), one workgroup per (batch, head), no barrier after the load.
bool attn_res_ok(const AttnP& p);
int attn_res_fwd(const AttnP& p, int mode, hipStream_t st);
int attn_res_bwd(const AttnP& p, int mode, hipStream_t st);       // dq and dk/dv kernels (delta already computed)
