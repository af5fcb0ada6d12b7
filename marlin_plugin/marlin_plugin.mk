# Include from Marlin's Makefile AFTER `include $(FRAMEWORK_DIR)/app.mk`:
#     MARLIN_HIP_DIR ?= /path/to/this/repo
#     include $(MARLIN_HIP_DIR)/marlin_plugin/marlin_plugin.mk
# Adds the shim classes (HipAdamsBashforthMoulton, HipFFTMechanics) to the application and links libmarlin_hip.so.
# Nothing here is built by this repository: the classes need MOOSE / libMesh / libTorch headers (MOOSE_DIR).
ifneq ($(wildcard $(MOOSE_DIR)/framework/include/base/MooseObject.h),)
MARLIN_HIP_DIR      ?= $(abspath $(dir $(lastword $(MAKEFILE_LIST)))/..)
ADDITIONAL_INCLUDES += -I$(MARLIN_HIP_DIR)/include -I$(MARLIN_HIP_DIR)/marlin_plugin/include
ADDITIONAL_LIBS     += -L$(MARLIN_HIP_DIR)/marlin_amd/lib -lmarlin_hip -Wl,-rpath,$(MARLIN_HIP_DIR)/marlin_amd/lib
ADDITIONAL_SRC_DIRS += $(MARLIN_HIP_DIR)/marlin_plugin/src
else
$(info marlin_plugin.mk: MOOSE_DIR not set or not a MOOSE checkout -- the libmarlin_hip shim classes are not built)
endif
