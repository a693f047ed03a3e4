// shim: esekfom.hpp includes <boost/bind.hpp> but the code paths the harness instantiates never call boost::bind
#pragma once
