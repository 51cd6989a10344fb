// ORACLE / TEST INFRASTRUCTURE ONLY: boost::shared_ptr for the reference sources compiled in place (oracle/Makefile, target `ref`)
#pragma once
#include <memory>
namespace boost { template <typename T> using shared_ptr = std::shared_ptr<T>; }
