#define FVSRN_CD 4
#define FVSRN_PART 1
#include "kernels_inst.inc"
