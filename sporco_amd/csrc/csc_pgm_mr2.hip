// csc_pgm_mr2.hip -- csc_pgm_mr.hip for the heights 16 x {21 ... 30} (second translation unit).
#define SA_MR_PART 1
#include "csc_pgm_mr.hip"
