#define FVSRN_CD 3
#define FVSRN_PART 0
#include "kernels_inst.inc"
