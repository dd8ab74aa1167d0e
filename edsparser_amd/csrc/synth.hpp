#pragma once
#include "dev_util.hpp"
namespace edsx {
size_t synth_size(u32 S, u64 ncols);
size_t synth_size_aligned(u32 S, u64 ncols, u32 align);      // align <= 1: the plain image
void synth_generate(uint8_t* d_out, u64* d_desc, u32 S, u64 col0, u64 ncols, double v, u64 seed, hipStream_t st, u32 align = 0);
}
