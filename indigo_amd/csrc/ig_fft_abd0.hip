#define IG_ABD_PART 0
#include "ig_fft_abd_part.inc"
