#define FVSRN_CD 4
#define FVSRN_PART 5
#include "kernels_inst.inc"
