#include "../../ref_shim5/pangolin/pangolin.h"
