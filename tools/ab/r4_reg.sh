timeout 900 python tools/dbg/qt_regimes.py
