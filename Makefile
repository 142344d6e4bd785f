# Builds the gfx950 HIP library in-tree (the .so travels to the GPU box with the snapshot).
HIPCC ?= /opt/rocm/bin/hipcc
ARCH  ?= gfx950
CSRC  := ufvideo_amd/csrc
SRCS  := $(CSRC)/gemm.hip $(CSRC)/gemm256.hip $(CSRC)/gemm256_b.hip $(CSRC)/gemm256_q.hip $(CSRC)/gemm256_s.hip $(CSRC)/gemm256_r.hip $(CSRC)/gemm256_m.hip $(CSRC)/gemm256_m2.hip $(CSRC)/gemm_state.hip $(CSRC)/attn.hip $(CSRC)/ops.hip $(CSRC)/qwen2_decode.hip $(CSRC)/sam_heads.hip $(CSRC)/quant.hip $(CSRC)/loss.hip $(CSRC)/resize.hip $(CSRC)/train.hip $(CSRC)/sample.hip $(CSRC)/train_proj.hip $(CSRC)/attn_bwd.hip $(CSRC)/seg_train.hip $(CSRC)/stages.hip
OBJS  := $(SRCS:.hip=.o)
LIB   := ufvideo_amd/libufv_hip.so
FLAGS := --offload-arch=$(ARCH) -O3 -fPIC -std=c++17 -Wall -Wno-unused-function -Iinclude

all: $(LIB)

$(CSRC)/%.o: $(CSRC)/%.hip $(CSRC)/common.h $(CSRC)/gemm_epi.h $(CSRC)/gemm256_kernel.h $(CSRC)/gemm_state.h $(CSRC)/attn_vit.inc $(CSRC)/attn_vit_p2.inc $(CSRC)/attn_vit_p2_asm.inc $(CSRC)/attn_c128.inc $(CSRC)/attn_c128_asm.inc include/ufv.h
	$(HIPCC) $(FLAGS) -c $< -o $@

$(LIB): $(OBJS)
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -o $@ $(OBJS)

clean:
	rm -f $(OBJS) $(LIB)

.PHONY: all clean
