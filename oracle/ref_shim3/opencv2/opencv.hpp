#include "../lvref_cv3.hpp"
