#define FVSRN_CD 5
#define FVSRN_PART 5
#include "kernels_inst.inc"
