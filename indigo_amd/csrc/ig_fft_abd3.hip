#define IG_ABD_PART 3
#include "ig_fft_abd_part.inc"
