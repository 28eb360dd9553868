// csc_rows_mr2.hip -- the mixed-radix row kernels of csc_rows_mr.hip for the lengths N1 = 21 ... 30
// (a second translation unit: the two halves compile side by side).
#define SA_MR_PART 1
#include "csc_rows_mr.hip"
