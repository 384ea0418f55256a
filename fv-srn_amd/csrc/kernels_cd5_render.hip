#define FVSRN_CD 5
#define FVSRN_PART 1
#include "kernels_inst.inc"
