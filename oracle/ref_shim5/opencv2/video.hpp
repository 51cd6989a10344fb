#include "../lvref_main.hpp"
