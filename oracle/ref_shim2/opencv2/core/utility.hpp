#include "../lvref_cv_fs.hpp"
