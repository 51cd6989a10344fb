#include "ceres.h"
