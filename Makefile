# Builds longtail_amd/liblongtail_hip.so (HIP kernels for gfx950 + the plain-C plugin layer), in-tree so that
# the built library travels with the repository snapshot to the GPU box.
#   make            product library
#   make ablations  build/ablations/liblongtail_hip.so: the same sources with the earlier kernel formulations and debug switches
#   make oracle     test infrastructure (oracle/liblongtail_oracle.so and, when /root/reference exists, oracle/_ref)
#   make all        both
HIPCC   ?= /opt/rocm/bin/hipcc
CC      ?= gcc
ARCH    ?= gfx950
CSRC    := longtail_amd/csrc
OBJDIR  := build/obj
LIB     := longtail_amd/liblongtail_hip.so

HIP_SRC := $(CSRC)/lthip_ctx.hip $(CSRC)/k_buzhash.hip $(CSRC)/k_blake3.hip $(CSRC)/k_lz4.hip $(CSRC)/k_lz4_decode.hip $(CSRC)/k_zstd.hip $(CSRC)/k_zstd_decode.hip \
           $(CSRC)/k_dedup.hip $(CSRC)/k_gather.hip $(CSRC)/k_synth.hip $(CSRC)/version_index.hip $(CSRC)/ingest.hip $(CSRC)/comm.hip
C_SRC   := $(CSRC)/plugin/plugin_common.c $(CSRC)/plugin/plugin_chunker.c $(CSRC)/plugin/plugin_hash.c \
           $(CSRC)/plugin/plugin_codec.c $(CSRC)/plugin/plugin_codec_batch.c $(CSRC)/plugin/plugin_batch.c $(CSRC)/plugin/build_id.c $(CSRC)/plugin/partition.c
HIP_OBJ := $(patsubst $(CSRC)/%.hip,$(OBJDIR)/%.o,$(HIP_SRC))
C_OBJ   := $(patsubst $(CSRC)/plugin/%.c,$(OBJDIR)/%.o,$(C_SRC))

GENDIR  := build/gen
# header dependencies come from the compilers (-MMD -MP -> build/obj/*.d), never from a hand-written list: a header edit
# rebuilds exactly the objects that include it
HIPFLAGS := --offload-arch=$(ARCH) -O3 -std=c++17 -fPIC -fvisibility=hidden -Wall -Wno-unused-function -Iinclude -MMD -MP $(EXTRA_HIPFLAGS)
CFLAGS   := -O2 -std=c99 -D_POSIX_C_SOURCE=200809L -fPIC -fvisibility=hidden -Wall -Wextra -pthread -Iinclude -I$(GENDIR) -MMD -MP

# identity of the source tree (tools/build_id.py), refreshed on every make run; the stamp only changes when a source does
BUILD_ID := $(shell python3 tools/build_id.py --stamp $(GENDIR)/build_id.h)

.PHONY: lib oracle all clean prof ablations
lib: $(LIB)
all: lib oracle ablations

$(OBJDIR):
	mkdir -p $(OBJDIR)

$(OBJDIR)/%.o: $(CSRC)/%.hip | $(OBJDIR)
	$(HIPCC) $(HIPFLAGS) -c $< -o $@

$(OBJDIR)/%.o: $(CSRC)/plugin/%.c | $(OBJDIR)
	$(CC) $(CFLAGS) -c $< -o $@

-include $(HIP_OBJ:.o=.d) $(C_OBJ:.o=.d)

$(LIB): $(HIP_OBJ) $(C_OBJ)
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -o $@ $(HIP_OBJ) $(C_OBJ) -lpthread

oracle:
	$(MAKE) -C oracle

# The ABLATION build: the same sources with -DLTHIP_ABLATIONS -> build/ablations/liblongtail_hip.so.  It adds what the product library
# leaves out: earlier formulations of the kernels ($(CSRC)/ablations/*.inc) and debug paths, selected by LTHIP_* switches that the
# product does not read (lthip_internal.h: LTHIP_ABLATION_ENV).  The differential tests load it next to the product library
# (longtail_amd.lib.load_ablations, the gpu_abl fixture), the A/B tools through LTHIP_LIB_PATH.
ABLDIR   := build/ablations
ABL_HIP_OBJ := $(patsubst $(CSRC)/%.hip,$(ABLDIR)/%.o,$(HIP_SRC))
ABL_LIB  := $(ABLDIR)/liblongtail_hip.so
$(ABLDIR):
	mkdir -p $(ABLDIR)
$(ABLDIR)/%.o: $(CSRC)/%.hip | $(ABLDIR)
	$(HIPCC) $(HIPFLAGS) -DLTHIP_ABLATIONS -c $< -o $@
-include $(ABL_HIP_OBJ:.o=.d)
$(ABL_LIB): $(ABL_HIP_OBJ) $(C_OBJ)
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -o $@ $(ABL_HIP_OBJ) $(C_OBJ) -lpthread
ablations: $(ABL_LIB)

# debug build for tools/zb_prof.sh: the zstd entropy kernel with per-phase cycle counters (-DLTHIP_ZB_PROF)
prof: $(LIB)
	mkdir -p build/prof
	$(HIPCC) $(HIPFLAGS) -DLTHIP_ABLATIONS -DLTHIP_ZB_PROF -c $(CSRC)/k_zstd.hip -o build/prof/k_zstd.o
	$(HIPCC) $(HIPFLAGS) -DLTHIP_ABLATIONS -DLTHIP_ZB_PROF -c $(CSRC)/k_zstd_decode.hip -o build/prof/k_zstd_decode.o
	$(HIPCC) $(HIPFLAGS) -DLTHIP_ABLATIONS -DLTHIP_DEC_PROF -c $(CSRC)/k_lz4_decode.hip -o build/prof/k_lz4_decode.o
	$(HIPCC) $(HIPFLAGS) -DLTHIP_K5_PROF -c $(CSRC)/k_lz4.hip -o build/prof/k_lz4.o
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -o build/prof/liblongtail_hip_prof.so $(filter-out $(OBJDIR)/k_zstd.o $(OBJDIR)/k_zstd_decode.o $(OBJDIR)/k_lz4_decode.o $(OBJDIR)/k_lz4.o,$(HIP_OBJ)) build/prof/k_zstd.o build/prof/k_zstd_decode.o build/prof/k_lz4_decode.o build/prof/k_lz4.o $(C_OBJ) -lpthread

clean:
	rm -rf build $(LIB)
	$(MAKE) -C oracle clean
