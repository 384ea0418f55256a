#define FVSRN_CD 3
#define FVSRN_PART 2
#include "kernels_inst.inc"
