LLPF_LIB=$PWD/lib_steptiming.so timeout 120 python tools/dbg/qt_phases.py 2>&1 | head -30
