// K1's instances for pathtrace: see vpt_k1_instances.hip.h
#define VPT_INSTANCES_TU
#include "vpt_k1_instances.hip.h"
VPT_K1_SPLIT_INSTANCES(VPT_K1_DEFINE, K_PATH)
