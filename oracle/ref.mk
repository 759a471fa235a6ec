# oracle/ref.mk — builds oracle/_ref/liberasor_ref.so: the reference's OWN hot-path sources, compiled
# UNMODIFIED from where they lie under /root/reference, against the stand-in headers in oracle/stubs/
# (ros / pcl / Eigen / tf are not installed; their arithmetic delegates to oracle/third_party_restated.h).
# NOT A REFERENCE BUILD in the task's sense (the reference needs ROS / PCL / Eigen / tf, which this image lacks: unbuildable here) -- a
# CROSS-CHECK of the oracle's restatement of the reference-owned logic, see the header of erasor_oracle.cpp.
# TEST INFRASTRUCTURE ONLY: tests/test_oracle_vs_ref.py compares oracle/erasor_oracle.cpp with this library,
# bench.py times it beside the oracle (cpu_reference_sources; cpu_baseline is the oracle port).  Outputs go to oracle/_ref/ only (git-ignored,
# travels to the GPU box as a prebuilt file; /root/reference does not exist there).
# Flags follow the reference's build: no -march (CMakeLists.txt:4), so no FMA contraction.
# -fno-gnu-unique: oracle/ref.py loads one private copy of the library per object; GNU_UNIQUE symbols (statics of
# inline functions, e.g. the stub's topic registry) would otherwise be shared between live copies.
REF      ?= /root/reference
CXX      ?= g++
CXXFLAGS ?= -O2 -std=c++17 -ffp-contract=off -fno-gnu-unique -fPIC -w
INC       = -I$(REF)/include -I$(REF)/src/mapgen -Istubs
OUT       = _ref
REFSRC    = $(REF)/src/offline_map_updater/src
OBJS      = $(OUT)/erasor.o $(OUT)/erasor_utils.o $(OUT)/OfflineMapUpdater.o $(OUT)/ref_driver.o
STUBS     = stubs/ref_stubs.h third_party_restated.h

all: $(OUT)/liberasor_ref.so

$(OUT)/%.o: $(REFSRC)/%.cpp $(STUBS)
	@mkdir -p $(OUT)
	$(CXX) $(CXXFLAGS) $(INC) -c $< -o $@

# the driver reads private members the reference only exposes through RViz topics
$(OUT)/ref_driver.o: ref_driver.cpp $(STUBS) ../include/erasor_hip.h
	@mkdir -p $(OUT)
	$(CXX) $(CXXFLAGS) -fno-access-control $(INC) -c ref_driver.cpp -o $@

$(OUT)/liberasor_ref.so: $(OBJS)
	$(CXX) -shared -o $@ $(OBJS)

clean:
	rm -rf $(OUT)
