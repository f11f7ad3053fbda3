#pragma once  // torch is not used by the src/ kernels
