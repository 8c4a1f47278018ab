// declared-interface stand-in (tests/parthenon_iface/parthenon_iface.hpp): NOT Parthenon
#include "../parthenon_iface.hpp"
