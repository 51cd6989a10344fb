#include "../lvref_main3.hpp"
