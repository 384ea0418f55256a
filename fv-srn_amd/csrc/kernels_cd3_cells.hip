#define FVSRN_CD 3
#define FVSRN_PART 5
#include "kernels_inst.inc"
