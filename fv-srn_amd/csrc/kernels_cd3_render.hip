#define FVSRN_CD 3
#define FVSRN_PART 1
#include "kernels_inst.inc"
