#pragma once
// same enumerators as the reference (include/linearMpcHumanoid/general/Task.hpp:3-13)
enum class Task { Stand, Walk, Jump };
enum class SupportFoot { Right, Left, Double };
