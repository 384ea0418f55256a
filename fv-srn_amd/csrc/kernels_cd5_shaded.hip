#define FVSRN_CD 5
#define FVSRN_PART 2
#include "kernels_inst.inc"
