#define FVSRN_CD 1
#define FVSRN_PART 0
#include "kernels_inst.inc"
