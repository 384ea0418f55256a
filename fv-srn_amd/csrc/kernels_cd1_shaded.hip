#define FVSRN_CD 1
#define FVSRN_PART 2
#include "kernels_inst.inc"
