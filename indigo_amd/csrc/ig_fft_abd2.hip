#define IG_ABD_PART 2
#include "ig_fft_abd_part.inc"
