#pragma once  // cg::reduce is never called
