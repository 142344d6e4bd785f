# Builds the gfx950 HIP library in-tree (the .so travels to the GPU box with the snapshot).
#   make            the product library: -fvisibility=hidden, only the UFV_API entry points of include/ufv.h are dynamic symbols
#   make EXTRA=-DUFV_LAB_KERNELS   lab build: the diagnostic attention kernel ids too (tools/lab/build_variant_lib.sh)
#   make asan       host-side sanitizer build + run of the library's host logic (tests/asan/: cost model, split-K ring, error paths) -- CPU only
HIPCC ?= /opt/rocm/bin/hipcc
ARCH  ?= gfx950
CSRC  := ufvideo_amd/csrc
SRCS  := $(CSRC)/gemm.hip $(CSRC)/gemm256.hip $(CSRC)/gemm256_b.hip $(CSRC)/gemm256_q.hip $(CSRC)/gemm256_s.hip $(CSRC)/gemm256_r.hip $(CSRC)/gemm256_m.hip $(CSRC)/gemm256_m2.hip $(CSRC)/gemm_state.hip $(CSRC)/attn.hip $(CSRC)/ops.hip $(CSRC)/qwen2_decode.hip $(CSRC)/sam_heads.hip $(CSRC)/quant.hip $(CSRC)/loss.hip $(CSRC)/resize.hip $(CSRC)/train.hip $(CSRC)/sample.hip $(CSRC)/train_proj.hip $(CSRC)/attn_bwd.hip $(CSRC)/seg_train.hip $(CSRC)/stages.hip
OBJS  := $(SRCS:.hip=.o)
DEPS  := $(SRCS:.hip=.d)
LIB   := ufvideo_amd/libufv_hip.so
EXTRA ?=
FLAGS := --offload-arch=$(ARCH) -O3 -fPIC -std=c++17 -fvisibility=hidden -Wall -Wno-unused-function -Iinclude $(EXTRA)

all: $(LIB)

# header dependencies are the compiler's own (-MMD): a generated attention body rebuilds attn.o only
$(CSRC)/%.o: $(CSRC)/%.hip Makefile
	$(HIPCC) $(FLAGS) -MMD -MP -MF $(CSRC)/$*.d -c $< -o $@

$(LIB): $(OBJS) $(CSRC)/ufv_exports.map
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -fvisibility=hidden -Wl,--version-script=$(CSRC)/ufv_exports.map -o $@ $(OBJS)

-include $(DEPS)

clean:
	rm -f $(OBJS) $(DEPS) $(LIB)

.PHONY: all clean
