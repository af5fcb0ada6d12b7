// forwards to the stand-ins of tests/moose_stub/moose_stub.h (test infrastructure; see there)
#pragma once
#include "moose_stub.h"
