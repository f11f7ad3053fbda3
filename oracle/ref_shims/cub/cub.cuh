#pragma once
// CUB -> hipCUB (same API; rocPRIM back-end).  aggregator_impl.cu:122,142-145,193,219-224.
#include <hipcub/hipcub.hpp>
namespace cub = hipcub;
