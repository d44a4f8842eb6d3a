# Top-level build: libbdx.so (HIP, gfx950), the breakdancer-max CLI and the CPU oracle.
HIPCC ?= /opt/rocm/bin/hipcc
ARCH ?= gfx950
CSRC := breakdancer_amd/csrc
HOST := breakdancer_amd/host
HIPFLAGS := --offload-arch=$(ARCH) -O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-result -Iinclude
KERNELS := $(CSRC)/k1_classify.hip $(CSRC)/k2_compact.hip $(CSRC)/k3_regions.hip $(CSRC)/k4_join.hip $(CSRC)/k5_poisson.hip $(CSRC)/k6_assemble.hip $(CSRC)/k7_exchange.hip $(CSRC)/k9_shard.hip $(CSRC)/kz_inflate.hip $(CSRC)/kb_records.hip $(CSRC)/kc_insert_stats.hip $(CSRC)/bdx_api.hip
OBJS := $(KERNELS:.hip=.o) $(CSRC)/bdx_walk.o $(CSRC)/bdx_walk_reads.o
HOSTCOMMON := $(HOST)/options.cpp $(HOST)/config.cpp $(HOST)/bam_reader.cpp $(HOST)/fast_inflate.cpp $(HOST)/column_reader.cpp $(HOST)/producer.cpp $(HOST)/dumps.cpp $(HOST)/cache.cpp

all: breakdancer_amd/libbdx.so bin/breakdancer-max bin/bdx-dump-reads bin/bam2cfg bin/bdx-inflate-check bin/bdx-feed-probe oracle

$(CSRC)/%.o: $(CSRC)/%.hip $(wildcard $(CSRC)/*.h) include/bdx.h
	$(HIPCC) $(HIPFLAGS) -c $< -o $@

$(CSRC)/bdx_walk.o: $(CSRC)/bdx_walk.cpp $(CSRC)/bdx_walk.h include/bdx.h
	$(HIPCC) $(HIPFLAGS) -x hip -c $< -o $@

$(CSRC)/bdx_walk_reads.o: $(CSRC)/bdx_walk_reads.cpp $(CSRC)/bdx_walk.h $(CSRC)/bdx_dev.h include/bdx.h
	$(HIPCC) $(HIPFLAGS) -x hip -c $< -o $@

breakdancer_amd/libbdx.so: $(OBJS)
	$(HIPCC) --offload-arch=$(ARCH) -shared -o $@ $(OBJS)

HOSTFLAGS := -O2 -std=c++17 -ffp-contract=off -Wall -Iinclude -pthread

bin/breakdancer-max: $(HOSTCOMMON) $(HOST)/main.cpp $(wildcard $(HOST)/*.h) breakdancer_amd/libbdx.so
	@mkdir -p bin
	g++ $(HOSTFLAGS) -o $@ $(HOSTCOMMON) $(HOST)/main.cpp -Lbreakdancer_amd -lbdx -lz -Wl,-rpath,'$$ORIGIN/../breakdancer_amd' -Wl,-rpath,/opt/rocm/lib

bin/bdx-inflate-check: $(HOST)/inflate_check_main.cpp $(HOST)/fast_inflate.cpp $(HOST)/fast_inflate.h
	@mkdir -p bin
	g++ $(HOSTFLAGS) -O3 -o $@ $(HOST)/inflate_check_main.cpp $(HOST)/fast_inflate.cpp -lz

# measurement tool (bench.py): the feeder's ceilings -- page cache -> pinned -> HBM
bin/bdx-feed-probe: tools/feed_probe.hip
	@mkdir -p bin
	$(HIPCC) --offload-arch=$(ARCH) -O2 -std=c++17 -mavx2 -o $@ $< -lpthread

bin/bam2cfg: $(HOST)/bam2cfg_main.cpp $(HOST)/bam_reader.cpp $(HOST)/bam_reader.h breakdancer_amd/libbdx.so
	@mkdir -p bin
	g++ $(HOSTFLAGS) -o $@ $(HOST)/bam2cfg_main.cpp $(HOST)/bam_reader.cpp -Lbreakdancer_amd -lbdx -lz -lpthread -Wl,-rpath,'$$ORIGIN/../breakdancer_amd' -Wl,-rpath,/opt/rocm/lib

bin/bdx-dump-reads: $(HOSTCOMMON) $(HOST)/dump_main.cpp $(wildcard $(HOST)/*.h) breakdancer_amd/libbdx.so
	@mkdir -p bin
	g++ $(HOSTFLAGS) -o $@ $(HOSTCOMMON) $(HOST)/dump_main.cpp -Lbreakdancer_amd -lbdx -lz -Wl,-rpath,'$$ORIGIN/../breakdancer_amd' -Wl,-rpath,/opt/rocm/lib

oracle:
	$(MAKE) -C oracle

clean:
	rm -f $(CSRC)/*.o breakdancer_amd/libbdx.so bin/breakdancer-max bin/bdx-dump-reads bin/bam2cfg bin/bdx-feed-probe
	$(MAKE) -C oracle clean

.PHONY: all oracle clean
