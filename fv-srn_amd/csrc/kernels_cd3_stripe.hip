#define FVSRN_CD 3
#define FVSRN_PART 3
#include "kernels_inst.inc"
