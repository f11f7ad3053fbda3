#pragma once  // ATen is not used by the kernels or their launchers
