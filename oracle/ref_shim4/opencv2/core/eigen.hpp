#include "../../lvref_sfm.hpp"
