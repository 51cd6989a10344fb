#include "../lvref_ceres.hpp"
