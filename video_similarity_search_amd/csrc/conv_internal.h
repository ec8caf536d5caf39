// entry points shared between the translation units of the convolution kernels (not part of the C ABI)
#pragma once
#include "common.h"
// conv_wino2.hip: variant 31 of slic_conv_gemm
int slic_wino2_full_rows(const SlicConvArgs* a);                    // real outputs per full block of 64 tiles; 0 = not uniform
int slic_conv_wino2_launch(const SlicConvArgs* a, hipStream_t st, int nfull, float* slab, int pieces);   // nfull < 0: one piece
size_t slic_conv_wino2_split_workspace_bytes(const SlicConvArgs* a, int nfull, int pieces);
