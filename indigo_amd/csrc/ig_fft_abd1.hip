#define IG_ABD_PART 1
#include "ig_fft_abd_part.inc"
