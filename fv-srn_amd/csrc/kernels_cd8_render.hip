#define FVSRN_CD 8
#define FVSRN_PART 1
#include "kernels_inst.inc"
